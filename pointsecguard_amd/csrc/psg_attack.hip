// Attack-loop arithmetic on gfx950: layout transposes, CE-on-log-probs gradient, the fused
// sign-step / L-inf projection update, and segmentation statistics.  All elementwise / tiny
// reductions: HBM-bound by construction (12-52 bytes per point), coalesced along the point index.
//
// Reference (paths relative to /root/reference):
//   PointNet/attacks/torchattacks/attacks/nontarget.py:26,34-39   NB_attack loss + update
//   PointNet/attacks/torchattacks/attacks/target.py:27,36-43      tar_NB_attack loss + update
//   PointNet/NB_nontarget_test_semseg.py:188-211                  acc / per-class I, U, seen
#include <string.h>

#include "psg_common.h"

namespace {

__global__ void to_point_major_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int N, size_t total)
{
    // one thread per (b, n): reads are coalesced along n for every channel
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        size_t b = t / N, n = t % N;
        for (int c = 0; c < C; ++c) dst[t * C + c] = src[(b * C + c) * N + n];
    }
}

__global__ void to_channel_major_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int N,
                                        size_t total)
{
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        size_t b = t / N, n = t % N;
        for (int c = 0; c < C; ++c) dst[(b * C + c) * N + n] = src[t * C + c];
    }
}

constexpr int MAXC = 32;

// 256 rows per workgroup staged through LDS: the rows of a block are one contiguous piece of logp / dlogp, so global
// accesses are whole cache lines (a thread per row reading its n_cls floats directly strides the lanes 52 bytes apart);
// the per-row arithmetic is unchanged.
__global__ __launch_bounds__(256) void ce_logp_grad_kernel(const float *__restrict__ logp, const int32_t *__restrict__ labels, int target,
                                                            int rows, int rows_active, int n_cls, float scale, float *__restrict__ dlogp,
                                                            float *__restrict__ cost)
{
    __shared__ float s_z[256 * MAXC];
    const int r0 = blockIdx.x * 256, nr = min(256, rows - r0);
    const float *src = logp + (size_t)r0 * n_cls;
    for (int i = threadIdx.x; i < nr * n_cls; i += 256) s_z[i] = src[i];
    __syncthreads();
    const int r = r0 + (int)threadIdx.x;
    float g[MAXC];
    if ((int)threadIdx.x < nr) {
        if (r >= rows_active) {
            for (int c = 0; c < n_cls; ++c) g[c] = 0.0f;
        } else {
            const float *lp = s_z + threadIdx.x * n_cls;        // (n_cls = 13: an odd stride, conflict-free)
            float z[MAXC];
            float m = -INFINITY;
            for (int c = 0; c < n_cls; ++c) { z[c] = lp[c]; m = fmaxf(m, z[c]); }
            float s = 0.0f;
            for (int c = 0; c < n_cls; ++c) s += expf(z[c] - m);
            const float lse = logf(s);
            const int y = labels ? labels[r] : target;
            for (int c = 0; c < n_cls; ++c) {
                float lp2 = (z[c] - m) - lse;  // second log_softmax (CrossEntropyLoss applied to log-probs)
                float p = expf(lp2);
                g[c] = (p - (c == y ? 1.0f : 0.0f)) * scale;
                if (cost && c == y) atomicAdd(cost, -lp2 * scale);
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nr) {
        float *o = s_z + threadIdx.x * n_cls;
        for (int c = 0; c < n_cls; ++c) o[c] = g[c];
    }
    __syncthreads();
    float *dst = dlogp + (size_t)r0 * n_cls;
    for (int i = threadIdx.x; i < nr * n_cls; i += 256) dst[i] = s_z[i];
}

__global__ void pgd_step_kernel(float *__restrict__ x, const float *__restrict__ grad, const float *__restrict__ ori,
                                const uint8_t *__restrict__ mask, int N, size_t total, float step, float eps, int last)
{
    // one thread per (room, point, colour channel); `step` = dir*alpha already rounded to fp32
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total * 3; t += (size_t)gridDim.x * blockDim.x) {
        size_t pt = t / 3;
        int ch = (int)(t % 3);
        if (mask && !mask[pt % N]) continue;
        size_t xi = pt * 9 + 3 + ch;
        float g = grad[xi];
        float sg = g > 0.0f ? 1.0f : (g < 0.0f ? -1.0f : 0.0f);
        float stepped = __fadd_rn(x[xi], __fmul_rn(step, sg));
        float o = ori[t];
        float eta = fminf(fmaxf(__fsub_rn(stepped, o), -eps), eps);
        float proj = fminf(fmaxf(__fadd_rn(o, eta), 0.0f), 1.0f);
        x[xi] = last ? stepped : proj;
    }
}

__global__ void seg_stats_kernel(const float *__restrict__ logp, const int32_t *__restrict__ labels, int rows,
                                 int n_cls, unsigned long long *__restrict__ counters, int32_t *__restrict__ pred_out)
{
    __shared__ unsigned int h[3 * MAXC];
    for (int i = threadIdx.x; i < 3 * MAXC; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        const float *lp = logp + (size_t)r * n_cls;
        float best = lp[0];
        int bi = 0;
        for (int c = 1; c < n_cls; ++c)
            if (lp[c] > best) { best = lp[c]; bi = c; }
        const int y = labels[r];
        if (pred_out) pred_out[r] = bi;
        atomicAdd(&h[y], 1u);                    // seen
        if (bi == y) {
            atomicAdd(&h[MAXC + y], 1u);         // intersection
            atomicAdd(&h[2 * MAXC + y], 1u);     // union
        } else {
            atomicAdd(&h[2 * MAXC + y], 1u);
            atomicAdd(&h[2 * MAXC + bi], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * n_cls; i += blockDim.x) {
        int k = i / n_cls, c = i % n_cls;
        unsigned v = h[k * MAXC + c];
        if (v) atomicAdd(&counters[k * n_cls + c], (unsigned long long)v);
    }
}

inline int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 2048); }

}  // namespace

extern "C" int psg_to_point_major(const float *src_cn, int B, int C, int N, float *dst_nc, psg_stream stream)
{
    PSG_REQUIRE(src_cn && dst_nc && B > 0 && C > 0 && N > 0, "psg_to_point_major: bad argument");
    size_t total = (size_t)B * N;
    hipLaunchKernelGGL(to_point_major_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src_cn, dst_nc, C,
                       N, total);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_to_channel_major(const float *src_nc, int B, int C, int N, float *dst_cn, psg_stream stream)
{
    PSG_REQUIRE(src_nc && dst_cn && B > 0 && C > 0 && N > 0, "psg_to_channel_major: bad argument");
    size_t total = (size_t)B * N;
    hipLaunchKernelGGL(to_channel_major_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src_nc, dst_cn,
                       C, N, total);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_ce_logp_grad(const float *logp, const int32_t *labels, int target, int rows, int rows_active,
                                int n_cls, float scale, float *dlogp_out, float *cost_out, psg_stream stream)
{
    PSG_REQUIRE(logp && dlogp_out && rows > 0, "psg_ce_logp_grad: bad argument");
    PSG_REQUIRE(n_cls > 0 && n_cls <= MAXC, "psg_ce_logp_grad: n_cls=%d out of range (1..%d)", n_cls, MAXC);
    PSG_REQUIRE(labels || (target >= 0 && target < n_cls), "psg_ce_logp_grad: target class %d out of range", target);
    PSG_REQUIRE(rows_active >= 0 && rows_active <= rows, "psg_ce_logp_grad: rows_active out of range");
    hipLaunchKernelGGL(ce_logp_grad_kernel, dim3(psg::ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream, logp,
                       labels, target, rows, rows_active, n_cls, scale, dlogp_out, cost_out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_pgd_step(float *x, const float *grad, const float *ori, const uint8_t *mask, int B, int N,
                            float alpha, float eps, float dir, int last, psg_stream stream)
{
    PSG_REQUIRE(x && grad && ori && B > 0 && N > 0, "psg_pgd_step: bad argument");
    size_t total = (size_t)B * N;
    hipLaunchKernelGGL(pgd_step_kernel, dim3(grid_for(total * 3)), dim3(256), 0, (hipStream_t)stream, x, grad, ori, mask,
                       N, total, dir * alpha, eps, last);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_seg_stats(const float *logp, const int32_t *labels, int rows, int n_cls, long long *counters,
                             int32_t *pred_out, psg_stream stream)
{
    PSG_REQUIRE(logp && labels && counters && rows > 0, "psg_seg_stats: bad argument");
    PSG_REQUIRE(n_cls > 0 && n_cls <= MAXC, "psg_seg_stats: n_cls=%d out of range (1..%d)", n_cls, MAXC);
    hipLaunchKernelGGL(seg_stats_kernel, dim3(std::min(256, psg::ceil_div(rows, 256))), dim3(256), 0,
                       (hipStream_t)stream, logp, labels, rows, n_cls, (unsigned long long *)counters, pred_out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// =============================================================================================
// Whole-scene evaluation harness: vote pool, vote arg-max statistics, L2 distance.
// Reference: PointNet/NB_nontarget_test_semseg.py:55-62 (add_vote), :184 (torch.dist), :219-241 (per-scene IoU).
// =============================================================================================
namespace {

// pool[point_idx[r]][argmax logp[r]] += 1 where weight[r] != 0 (integer votes: the adds commute, result exact)
__global__ void vote_add_kernel(const float *__restrict__ logp, const int32_t *__restrict__ pred_in,
                                const int32_t *__restrict__ point_idx, const float *__restrict__ weight, int rows,
                                int n_cls, int n_points, int32_t *__restrict__ pool, int *__restrict__ bad)
{
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        if (weight && weight[r] == 0.0f) continue;
        int bi;
        if (pred_in) bi = pred_in[r];
        else {
            const float *lp = logp + (size_t)r * n_cls;
            float best = lp[0];
            bi = 0;
            for (int c = 1; c < n_cls; ++c)
                if (lp[c] > best) { best = lp[c]; bi = c; }
        }
        const int p = point_idx[r];
        if (p < 0 || p >= n_points || bi < 0 || bi >= n_cls) { atomicOr(bad, 1); continue; }
        atomicAdd(&pool[(size_t)p * n_cls + bi], 1);
    }
}

// per scene: pred = argmax over the votes (first index on ties, np.argmax), then the seen / correct / union counters
__global__ void vote_stats_kernel(const int32_t *__restrict__ pool, const int32_t *__restrict__ labels, int n_points,
                                  int n_cls, unsigned long long *__restrict__ counters, int32_t *__restrict__ pred_out)
{
    __shared__ unsigned int h[3 * MAXC];
    for (int i = threadIdx.x; i < 3 * MAXC; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_points; p += gridDim.x * blockDim.x) {
        const int32_t *v = pool + (size_t)p * n_cls;
        int best = v[0], bi = 0;
        for (int c = 1; c < n_cls; ++c)
            if (v[c] > best) { best = v[c]; bi = c; }
        const int y = labels[p];
        if (pred_out) pred_out[p] = bi;
        atomicAdd(&h[y], 1u);
        if (bi == y) {
            atomicAdd(&h[MAXC + y], 1u);
            atomicAdd(&h[2 * MAXC + y], 1u);
        } else {
            atomicAdd(&h[2 * MAXC + y], 1u);
            atomicAdd(&h[2 * MAXC + bi], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * n_cls; i += blockDim.x) {
        int k = i / n_cls, c = i % n_cls;
        unsigned v = h[k * MAXC + c];
        if (v) atomicAdd(&counters[k * n_cls + c], (unsigned long long)v);
    }
}

// sum of squared differences in double (one partial per workgroup, fixed tree inside the workgroup; the final sum
// over <= 256 partials runs on one thread in index order): deterministic
__global__ void sqdiff_partial_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n,
                                      double *__restrict__ part)
{
    __shared__ double sh[256];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double d = (double)a[i] - (double)b[i];
        acc += d * d;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ void sqdiff_final_kernel(const double *__restrict__ part, int n_part, float *__restrict__ out)
{
    double acc = 0.0;
    for (int i = 0; i < n_part; ++i) acc += part[i];
    out[0] = (float)sqrt(acc);
}

}  // namespace

extern "C" int psg_vote_add(const float *logp, const int32_t *pred, const int32_t *point_idx, const float *weight,
                            int rows, int n_cls, int n_points, int32_t *pool, int32_t *bad_flag, psg_stream stream)
{
    PSG_REQUIRE((logp || pred) && point_idx && pool && bad_flag && rows > 0 && n_points > 0, "psg_vote_add: bad argument");
    PSG_REQUIRE(n_cls > 0 && n_cls <= MAXC, "psg_vote_add: n_cls=%d out of range (1..%d)", n_cls, MAXC);
    hipLaunchKernelGGL(vote_add_kernel, dim3(grid_for((size_t)rows)), dim3(256), 0, (hipStream_t)stream, logp, pred,
                       point_idx, weight, rows, n_cls, n_points, pool, bad_flag);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_vote_stats(const int32_t *pool, const int32_t *labels, int n_points, int n_cls, long long *counters,
                              int32_t *pred_out, psg_stream stream)
{
    PSG_REQUIRE(pool && labels && counters && n_points > 0, "psg_vote_stats: bad argument");
    PSG_REQUIRE(n_cls > 0 && n_cls <= MAXC, "psg_vote_stats: n_cls=%d out of range (1..%d)", n_cls, MAXC);
    hipLaunchKernelGGL(vote_stats_kernel, dim3(std::min(256, psg::ceil_div(n_points, 256))), dim3(256), 0,
                       (hipStream_t)stream, pool, labels, n_points, n_cls, (unsigned long long *)counters, pred_out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_l2_dist(const float *a, const float *b, size_t n, double *scratch256, float *out, psg_stream stream)
{
    PSG_REQUIRE(a && b && scratch256 && out && n > 0, "psg_l2_dist: bad argument");
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 256);
    hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, n, scratch256);
    PSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sqdiff_final_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scratch256, blocks, out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// =============================================================================================
// NU (norm-unbounded, Adam in tanh space) attack arithmetic.
// Reference: PointNet/attacks/torchattacks/attacks/nontarget.py:52-135 (NU_attack),
//            PointNet/attacks/torchattacks/attacks/target.py:62-175 (tar_NU_attack).
// =============================================================================================
namespace {

// w = atanh(2c - 1) written the reference's way: 0.5*log((1+x)/(1-x))  (nontarget.py:110-116)
__global__ void nu_inverse_tanh_kernel(const float *__restrict__ x0, float *__restrict__ w, size_t rows)
{
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < rows * 3; t += (size_t)gridDim.x * blockDim.x) {
        float c = x0[(t / 3) * 9 + 3 + (t % 3)];
        float x = __fsub_rn(__fmul_rn(c, 2.0f), 1.0f);
        w[t] = __fmul_rn(0.5f, logf(__fdiv_rn(__fadd_rn(1.0f, x), __fsub_rn(1.0f, x))));
    }
}

// colour = 1/2 * (tanh(w) + 1)  (nontarget.py:107-108) written into channels 3:6 of x0, masked points only
__global__ void nu_tanh_color_kernel(const float *__restrict__ w, const uint8_t *__restrict__ mask, float *__restrict__ x0,
                                     int N, size_t rows, int mask_per_room)
{
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < rows * 3; t += (size_t)gridDim.x * blockDim.x) {
        size_t pt = t / 3;
        if (mask && !mask[mask_per_room ? pt : pt % N]) continue;
        x0[pt * 9 + 3 + (t % 3)] = __fmul_rn(0.5f, __fadd_rn(tanhf(w[t]), 1.0f));
    }
}

// f-loss of the paper on softmax(log-probs) + its gradient w.r.t. the log-probs (nontarget.py:119-128,
// target.py:148-168): f = clamp(tsign * (p_y - max_{k != y} p_k), min = -kappa), summed over points.
__global__ void nu_f_loss_grad_kernel(const float *__restrict__ logp, const int32_t *__restrict__ labels, int target,
                                      int rows, int n_cls, float kappa, float tsign, float *__restrict__ dlogp,
                                      float *__restrict__ f_sum, int32_t *__restrict__ pred, int rows_per_sum)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    float fval = 0.0f;
    if (r < rows) {
        const float *lp = logp + (size_t)r * n_cls;
        float p[MAXC];
        float m = -INFINITY;
        int am = 0;
        for (int c = 0; c < n_cls; ++c) {
            p[c] = lp[c];
            if (p[c] > m) { m = p[c]; am = c; }
        }
        if (pred) pred[r] = am;
        float s = 0.0f;
        for (int c = 0; c < n_cls; ++c) { p[c] = expf(p[c] - m); s += p[c]; }
        for (int c = 0; c < n_cls; ++c) p[c] = p[c] / s;
        const int y = labels ? labels[r] : target;
        float oth = -1.0f;
        int oi = 0;
        for (int c = 0; c < n_cls; ++c)
            if (c != y && p[c] > oth) { oth = p[c]; oi = c; }
        if (oth < 0.0f) oth = 0.0f;  // (1 - onehot) * p is 0 at the true class: the max is never below 0
        const float val = tsign * (p[y] - oth);
        const bool pass = val >= -kappa;
        fval = pass ? val : -kappa;
        const float gy = pass ? tsign : 0.0f, go = pass ? -tsign : 0.0f;
        const float dot = gy * p[y] + go * p[oi];
        float *g = dlogp + (size_t)r * n_cls;
        for (int c = 0; c < n_cls; ++c) {
            float gc = c == y ? gy : (c == oi ? go : 0.0f);
            g[c] = p[c] * (gc - dot);
        }
    }
    // reduction of the f values (rows_per_sum > 0: one sum per room, a multiple of 64 rows each).  One atomic per
    // workgroup when all its rows belong to one sum, else one per wave: thousands of waves adding into a handful of
    // addresses serialise on them (the per-room launches spent most of their time there).
    for (int o = 32; o >= 1; o >>= 1) fval += __shfl_xor(fval, o);
    if (!f_sum) return;
    if (rows_per_sum == 0 || rows_per_sum % (int)blockDim.x == 0) {
        __shared__ float s_part[16];
        const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
        if ((threadIdx.x & 63) == 0) s_part[wv] = fval;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.0f;
            for (int w = 0; w < nw; ++w) tot += s_part[w];
            const int r0 = blockIdx.x * blockDim.x;
            if (r0 < rows) atomicAdd(f_sum + (rows_per_sum ? r0 / rows_per_sum : 0), tot);
        }
    } else if ((threadIdx.x & 63) == 0 && r < rows) {
        atomicAdd(f_sum + r / rows_per_sum, fval);
    }
}

// f-loss of the ResGCN NU attacks, on raw logits (ResGCN/.../attacks/colper.py:108-113, tcolper.py:145-163).
// The reference multiplies the logits by one-hot masks before torch.max, so a masked-out slot contributes a 0:
//   mode 0 (NU_attack.f):        j = max(z_y, 0),            i = max(max_{k!=y} z_k, 0),  f = clamp(t*(j - i), -kappa)
//   mode 1 (tar_NU non_f):       j = z_y,                    i = max(max_{k!=y} z_k, 0),  f = clamp(t*(j - i), -kappa)
//   mode 2 (tar_NU tar_f, y=T):  j = max(max_{k!=T} z_k, 0), i = z_T,                     f = clamp(t*(j - i), -kappa)
// Modes 1/2 only count batch row 0 under `mask` ([0][self.mask] in the reference).  Writes scale * d(sum f)/dz.
__global__ void gcn_f_loss_grad_kernel(const float *__restrict__ z, const int32_t *__restrict__ labels, int target,
                                       const uint8_t *__restrict__ mask, int mode, int rows, int N, int n_cls, float kappa,
                                       float tsign, float scale, float *__restrict__ dz, float *__restrict__ f_sum,
                                       int32_t *__restrict__ pred)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    float fval = 0.0f;
    if (r < rows) {
        const float *zr = z + (size_t)r * n_cls;
        float v[MAXC];
        float m = -INFINITY;
        int am = 0;
        for (int c = 0; c < n_cls; ++c) {
            v[c] = zr[c];
            if (v[c] > m) { m = v[c]; am = c; }
        }
        if (pred) pred[r] = am;
        float *g = dz + (size_t)r * n_cls;
        for (int c = 0; c < n_cls; ++c) g[c] = 0.0f;
        const bool counted = mode == 0 || (r < N && (!mask || mask[r]));
        if (counted) {
            const int y = (mode == 2 || !labels) ? target : labels[r];
            float oth = 0.0f;       // the zeroed slot of the true class takes part in the max
            int oi = -1;
            for (int c = 0; c < n_cls; ++c)
                if (c != y && v[c] > oth) { oth = v[c]; oi = c; }
            float own = v[y];
            bool own_live = true;
            if (mode == 0 && !(own > 0.0f)) { own = 0.0f; own_live = false; }
            const float jv = mode == 2 ? oth : own, iv = mode == 2 ? own : oth;
            const float val = tsign * (jv - iv);
            const bool pass = val >= -kappa;
            fval = pass ? val : -kappa;
            if (pass) {
                const float gs = tsign * scale;
                if (mode == 2) {
                    if (oi >= 0) g[oi] += gs;
                    g[y] -= gs;
                } else {
                    if (own_live) g[y] += gs;
                    if (oi >= 0) g[oi] -= gs;
                }
            }
        }
    }
    for (int o = 32; o >= 1; o >>= 1) fval += __shfl_xor(fval, o);
    if ((threadIdx.x & 63) == 0 && f_sum) atomicAdd(f_sum, fval);
}

// Smooth loss (nontarget.py:131-135): for every adversarial colour of a room the `nb` smallest Euclidean distances to the
// reference colours; returns their sum and d(sum)/d(adv colour).
// A workgroup of 1024 threads stages the room's reference colours (+ |r|^2) in LDS once and serves 256 queries, FOUR lanes per
// query: each scans every 4th reference keeping its own sorted top-NBT in registers (a sorted insertion with static
// indexing; after the first few hundred references almost every candidate fails the one compare against the list's last
// entry), then the four short lists are merged nb times through two DPP-sized shuffles by (distance, index) - the lower
// index wins a tie, as a sequential scan with a strict '<' would - and the winner shifts its list.  Two workgroups (32
// waves) share a CU, and a 32-room lockstep launch (psg_smooth_knn_rooms) is exactly one round of 512 workgroups.
// (Round 2's version gave 16 lanes and a 64-KB staging copy to every 16 queries and merged in one lane: 2 workgroups =
// 8 waves per CU, 122 us for one room and 2.2 ms for 32 - half of the GPU time of a lockstep tar_NU step.)
constexpr int SM_MAX_NB = 16;
constexpr int SM_T = 1024;
constexpr int SM_REFRESH = 64;              // scan iterations between two updates of the query's common threshold
// SUB = lanes per query (4: 256 queries per workgroup, for launches of 16+ rooms; 16: 64 queries per workgroup, so that a
// one-room call still spreads over 64 workgroups)

// one step of the SUB-way merge: the smallest head of the query's SUB sorted lists by (distance, index) is returned in
// every lane of the query and popped from the one list that holds it (reference indices are unique across the lists;
// exhausted lists show (inf, 0x7FFFFFFF) and may pop together, which changes nothing)
template <int NBT, int SM_SUB>
__device__ __forceinline__ void smooth_pop_min(float (&bd)[NBT], int (&bi)[NBT], float &best, int &bidx)
{
    best = bd[0];
    bidx = bi[0];
#pragma unroll
    for (int o = 1; o < SM_SUB; o <<= 1) {
        const float od = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }
    }
    if (bi[0] == bidx) {
#pragma unroll
        for (int u = 0; u + 1 < NBT; ++u) { bd[u] = bd[u + 1]; bi[u] = bi[u + 1]; }
        bd[NBT - 1] = INFINITY; bi[NBT - 1] = 0x7FFFFFFF;
    }
}

typedef float sm_v2f __attribute__((ext_vector_type(2)));   // operand of the packed-fp32 pipe (v_pk_*_f32)

template <int NBT, int SM_SUB>
__global__ __launch_bounds__(SM_T) void smooth_knn_kernel(const float *__restrict__ adv, int adv_stride,
                                                          const float *__restrict__ ref, int ref_stride, int N, int nb,
                                                          float *__restrict__ dist_sum, float *__restrict__ grad, int symmetric,
                                                          size_t adv_room_stride, size_t ref_room_stride,
                                                          int32_t *__restrict__ nn_io, int have_prev)
{
    // blockIdx.y = room of a lockstep batch (psg_smooth_knn_rooms; a single launch of the one-room entry has one slice)
    adv += blockIdx.y * adv_room_stride;
    ref += blockIdx.y * ref_room_stride;
    grad += (size_t)blockIdx.y * N * 3;
    if (dist_sum) dist_sum += blockIdx.y;
    if (nn_io) nn_io += (size_t)blockIdx.y * N * nb;
    constexpr int SM_QPB = SM_T / SM_SUB;                   // queries per workgroup
    // Reference colours in LDS as four planes x, y, z, |r|^2, each split into SUB runs: run s holds the references
    // s, s + SUB, s + 2 SUB, .. (the ones lane s of a query scans) contiguously, so one ds_read_b128 per plane brings four of
    // them and two references share every instruction of the distance arithmetic on the packed-fp32 pipe.  Runs are 8 floats
    // apart beyond their length (the SUB lanes of a query read different runs at the same offset: different banks); slots
    // past N hold |r|^2 = inf and are never admitted.
    const int tps = ((N + SM_SUB - 1) / SM_SUB + 3) & ~3, run = tps + 8;
    extern __shared__ float s_pl[];
    float *s_x = s_pl, *s_y = s_pl + SM_SUB * run, *s_z = s_pl + 2 * SM_SUB * run, *s_q = s_pl + 3 * SM_SUB * run;
    for (int p = threadIdx.x; p < SM_SUB * tps; p += SM_T) {
        const int sr = p / tps, t = p - sr * tps, i = sr + SM_SUB * t;
        float x = 0.f, y = 0.f, z = 0.f, q = INFINITY;
        if (i < N) {
            x = ref[(size_t)i * ref_stride]; y = ref[(size_t)i * ref_stride + 1]; z = ref[(size_t)i * ref_stride + 2];
            q = x * x + y * y + z * z;
        }
        s_x[sr * run + t] = x; s_y[sr * run + t] = y; s_z[sr * run + t] = z; s_q[sr * run + t] = q;
    }
    __syncthreads();
    const int ql = threadIdx.x / SM_SUB, sub = threadIdx.x % SM_SUB;
    const int i = blockIdx.x * SM_QPB + ql;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (i < N) { ax = adv[(size_t)i * adv_stride]; ay = adv[(size_t)i * adv_stride + 1]; az = adv[(size_t)i * adv_stride + 2]; }
    float bd[NBT];
    int bi[NBT];
#pragma unroll
    for (int t = 0; t < NBT; ++t) { bd[t] = INFINITY; bi[t] = 0x7FFFFFFF; }
    // torch.cdist evaluates |a|^2 + |r|^2 - 2 a.r through a matmul (euclid_dist, clamp_min(0), sqrt): the
    // cancellation noise (~1e-7 in d^2, ~3e-4 in d) is part of the reference's loss surface -- it is what
    // keeps the gradient of a colour that has barely moved from its original near 0 instead of a unit
    // vector of rounding noise -- so the same expansion is used here (not bit-identical to MKL's order).
    const float asq = ax * ax + ay * ay + az * az;
    const sm_v2f m2x = {-2.0f * ax, -2.0f * ax}, m2y = {-2.0f * ay, -2.0f * ay}, m2z = {-2.0f * az, -2.0f * az}, asq2 = {asq, asq};
    // thr: a candidate is looked at only below min(this lane's worst kept distance, the nb-th smallest distance the
    // query's SUB lanes hold TOGETHER at the last refresh).  The common bound is what keeps the insertion branch rare:
    // a wave serves 64 / SUB queries, it runs the insertion network whenever ANY lane passes, and a lane's own
    // list only tightens as NBT / n.  Exact: everything kept at a refresh has a lower index than anything scanned later,
    // so a later candidate at exactly the bound loses the (distance, index) tie and '<' drops nothing that is wanted.
    // (The filter compares the distance BEFORE its clamp at 0: a negative one passes a positive bound either way, and
    // against a bound of 0 it only enters the insertion code, where the clamped value is refused like every other 0.)
    float thr = i < N ? INFINITY : -INFINITY;
    if (nn_io && have_prev) {
        // The optimiser moves a colour a little per step, so the nb references that were nearest one step ago are a sharp
        // and RIGOROUS start: the largest of their current distances bounds the nb-th smallest distance from above
        // (they are nb distinct references), and only the handful of references inside that ball ever reach the
        // insertion code.  The bound is taken a few ulps up so that the reference defining it passes the '<'.
        float m = -1.0f;
        bool ok = i < N;
        for (int t = sub; t < nb; t += SM_SUB) {
            const int jn = ok ? nn_io[(size_t)i * nb + t] : 0;
            if (jn < 0 || jn >= N) { ok = false; break; }
            const int pos = (jn % SM_SUB) * run + jn / SM_SUB;
            float d2 = __fmaf_rn(m2z[0], s_z[pos], __fmaf_rn(m2y[0], s_y[pos], __fmul_rn(m2x[0], s_x[pos])));
            d2 = __fadd_rn(__fadd_rn(d2, asq), s_q[pos]);
            m = fmaxf(m, fmaxf(d2, 0.0f));
        }
        unsigned bad = ok ? 0u : 1u;
#pragma unroll
        for (int o = 1; o < SM_SUB; o <<= 1) {
            m = fmaxf(m, __shfl_xor(m, o));
            bad |= (unsigned)__shfl_xor((int)bad, o);
        }
        if (!bad && i < N && m >= 0.0f) thr = m * 1.000001f + 1e-30f;
    }
    const float *px = s_x + sub * run, *py = s_y + sub * run, *pz = s_z + sub * run, *pq = s_q + sub * run;
    for (int t0 = 0; t0 < tps; t0 += SM_REFRESH) {
        const int t1 = t0 + SM_REFRESH < tps ? t0 + SM_REFRESH : tps;
        for (int t = t0; t < t1; t += 4) {
            const float4 X = *(const float4 *)(px + t), Y = *(const float4 *)(py + t), Z = *(const float4 *)(pz + t),
                         Q = *(const float4 *)(pq + t);
            // two references per instruction; per component exactly fma(m2z, z, fma(m2y, y, m2x * x)) then (+ asq) + q
            sm_v2f da = m2x * sm_v2f{X.x, X.y}, db = m2x * sm_v2f{X.z, X.w};
            da = __builtin_elementwise_fma(m2y, sm_v2f{Y.x, Y.y}, da); db = __builtin_elementwise_fma(m2y, sm_v2f{Y.z, Y.w}, db);
            da = __builtin_elementwise_fma(m2z, sm_v2f{Z.x, Z.y}, da); db = __builtin_elementwise_fma(m2z, sm_v2f{Z.z, Z.w}, db);
            da = (da + asq2) + sm_v2f{Q.x, Q.y};
            db = (db + asq2) + sm_v2f{Q.z, Q.w};
            const float d2[4] = {da[0], da[1], db[0], db[1]};
            if (fminf(fminf(d2[0], d2[1]), fminf(d2[2], d2[3])) < thr) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (d2[u] < thr) {
                        float cd = fmaxf(d2[u], 0.0f);
                        int ci = sub + SM_SUB * (t + u);
#pragma unroll
                        for (int k = 0; k < NBT; ++k) {   // sorted insertion with static indexing (arrays stay in registers)
                            if (cd < bd[k]) {
                                float td = bd[k]; int ti = bi[k];
                                bd[k] = cd; bi[k] = ci;
                                cd = td; ci = ti;
                            }
                        }
                        thr = fminf(thr, bd[NBT - 1]);
                    }
                }
            }
        }
        if (t1 < tps) {   // another chunk follows (uniform over the workgroup)
            float cdist[NBT];
            int cidx[NBT];
#pragma unroll
            for (int t = 0; t < NBT; ++t) { cdist[t] = bd[t]; cidx[t] = bi[t]; }
            float kth = INFINITY;
            int kidx;
            for (int t = 0; t < nb; ++t) smooth_pop_min<NBT, SM_SUB>(cdist, cidx, kth, kidx);
            thr = fminf(thr, kth);
        }
    }
    // SUB-way merge: nb times the smallest head by (distance, index); every lane of the query follows the same sequence
    float local = 0.0f, gx = 0.f, gy = 0.f, gz = 0.f;
    for (int t = 0; t < nb; ++t) {
        float best;
        int bidx;
        smooth_pop_min<NBT, SM_SUB>(bd, bi, best, bidx);
        if (bidx == 0x7FFFFFFF) break;   // fewer than nb references (uniform over the query's lanes)
        if (sub == 0 && i < N) {
            if (nn_io) nn_io[(size_t)i * nb + t] = bidx;          // next step's start
            const float d = sqrtf(best);
            local += d;
            if (d > 0.0f) {
                const int pos = (bidx % SM_SUB) * run + bidx / SM_SUB;
                const float ux = (ax - s_x[pos]) / d, uy = (ay - s_y[pos]) / d, uz = (az - s_z[pos]) / d;
                gx += ux; gy += uy; gz += uz;
                if (symmetric) {  // the neighbour is an adversarial colour too: it receives the opposite pull
                    atomicAdd(grad + (size_t)bidx * 3, -ux);
                    atomicAdd(grad + (size_t)bidx * 3 + 1, -uy);
                    atomicAdd(grad + (size_t)bidx * 3 + 2, -uz);
                }
            }
        }
    }
    if (sub == 0 && i < N) {
        if (symmetric) {
            atomicAdd(grad + (size_t)i * 3, gx); atomicAdd(grad + (size_t)i * 3 + 1, gy); atomicAdd(grad + (size_t)i * 3 + 2, gz);
        } else {
            grad[(size_t)i * 3] = gx; grad[(size_t)i * 3 + 1] = gy; grad[(size_t)i * 3 + 2] = gz;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && dist_sum) atomicAdd(dist_sum, local);
}

// Gradient assembly + torch.optim.Adam single-tensor update on w (fp32, torch's operation order).
// Set by psg_pn2_nu_window while it enqueues the steps of a window that is (or will be) replayed as a hipGraph: the Adam
// step and the latch of step i then read {step, lr / (1 - beta1^t), sqrt(1 - beta2^t)} from this device row instead of their
// arguments.  Null everywhere else.
static thread_local const float *g_nu_dconsts = nullptr;

__global__ void nu_adam_step_kernel(float *__restrict__ w, float *__restrict__ m, float *__restrict__ v,
                                    const uint8_t *__restrict__ mask, const float *__restrict__ dx0,
                                    const float *__restrict__ x0, const float *__restrict__ ori,
                                    const float *__restrict__ smooth_grad, float c_smooth, float c_l2, float beta1,
                                    float beta2, float eps, float step_size, float bc2_sqrt, int N, size_t rows,
                                    float *__restrict__ l2_sum, const uint8_t *__restrict__ room_active,
                                    const float *__restrict__ dconsts)
{
    // (a replayed hipGraph cannot change kernel arguments: the step-dependent constants of a captured window come from memory)
    if (dconsts) { step_size = dconsts[1]; bc2_sqrt = dconsts[2]; }
    // gridDim.y > 1: slice y is one room of a lockstep batch (rows = N) with its own mask row, smoothness gradient and
    // L2 sum; a room that has left its loop (room_active[y] == 0) is not touched.  gridDim.y == 1: the reference's batch
    // call (rows = B N, one mask, the smoothness gradient on batch row 0 only, one sum).
    if (gridDim.y > 1) {
        const size_t y = blockIdx.y, o3 = y * rows * 3;
        if (room_active && !room_active[y]) return;
        w += o3; m += o3; v += o3; ori += o3;
        dx0 += y * rows * 9; x0 += y * rows * 9;
        if (mask) mask += y * rows;
        if (smooth_grad) smooth_grad += o3;
        if (l2_sum) l2_sum += y;
    }
    float l2 = 0.0f;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < rows * 3; t += (size_t)gridDim.x * blockDim.x) {
        const size_t pt = t / 3;
        const int ch = (int)(t % 3);
        if (mask && !mask[pt % N]) continue;
        const float color = x0[pt * 9 + 3 + ch];
        const float diff = color - ori[t];
        l2 += diff * diff;
        float g = dx0[pt * 9 + 3 + ch] + c_l2 * 2.0f * diff;
        if (smooth_grad && pt < (size_t)N) g += c_smooth * smooth_grad[t];
        const float th = tanhf(w[t]);
        g = g * 0.5f * (1.0f - th * th);
        const float mm = __fadd_rn(m[t], __fmul_rn(__fsub_rn(g, m[t]), 1.0f - beta1));
        const float vv = __fadd_rn(__fmul_rn(v[t], beta2), __fmul_rn(1.0f - beta2, __fmul_rn(g, g)));
        m[t] = mm;
        v[t] = vv;
        const float denom = __fadd_rn(__fdiv_rn(sqrtf(vv), bc2_sqrt), eps);
        w[t] = __fadd_rn(w[t], __fmul_rn(-step_size, __fdiv_rn(mm, denom)));
    }
    for (int o = 32; o >= 1; o >>= 1) l2 += __shfl_xor(l2, o);
    if (!l2_sum) return;
    __shared__ float s_l2[4];                                  // (256 threads) one atomic per workgroup
    if ((threadIdx.x & 63) == 0) s_l2[threadIdx.x >> 6] = l2;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(l2_sum, (s_l2[0] + s_l2[1]) + (s_l2[2] + s_l2[3]));
}

}  // namespace

__device__ __forceinline__ unsigned wave_reduce_add_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_down((int)v, o);
    return v;            // lane 0 holds the sum
}

// ---- Per-step statistics and EXIT LATCH of the NU attacks on the device (round 4): what the reference evaluates on the
// host after every optimiser step (nontarget.py:87,95-96 `correct / 4096 < 1 / 13`; target.py:105-121 `target_acc`
// against 0.9 or 1 / 13) and what made the host loop wait for a read-back per step.  One workgroup per attack (a group of
// `rows` batch rows: rows = B for one call on a batch, rows = 1 for the lockstep rooms).  It counts the correct points and
// the target hits, files them with the step's f / Smooth / L2 sums (scal [3][G], zeroed here for the next step) into the
// history row hist_step [5][G], and evaluates the exit test in double precision exactly as the host did; the FIRST step
// whose test fires snapshots the group's image (x0 point-major -> out channel-major), records the step and clears the
// group's active byte (the lockstep optimiser skips inactive rooms).  The host then only reads the history where the
// reference itself has to touch the host: after step 0, after every 10th step (restart test + noise draw, lr halving)
// and at the end; a group that exits at step s returns its step-s image, at most 9 later steps are speculation.
__global__ __launch_bounds__(256) void nu_step_latch_kernel(const int32_t *__restrict__ pred, const int32_t *__restrict__ labels,
                                                             int target, const uint8_t *__restrict__ mask,
                                                             const int32_t *__restrict__ n_mask, int rows, int N, int mode,
                                                             float *__restrict__ scal, float *__restrict__ hist_step,
                                                             const float *__restrict__ x0, float *__restrict__ out,
                                                             uint8_t *__restrict__ active, int32_t *__restrict__ exit_step, int step,
                                                             int G, const float *__restrict__ dconsts)
{
    if (dconsts) step = __float_as_int(dconsts[0]);
    __shared__ unsigned s_cnt[2];
    __shared__ int s_exit;
    const int g = blockIdx.x, tid = threadIdx.x;
    if (tid < 2) s_cnt[tid] = 0u;
    __syncthreads();
    const size_t base = (size_t)g * rows * N;
    unsigned nc = 0, nt = 0;
    for (size_t i = tid; i < (size_t)rows * N; i += blockDim.x) {
        const int p = pred[base + i];
        const bool ok = p == labels[base + i];
        const bool in_mask = mask ? mask[(size_t)g * N + i % N] != 0 : true;
        nc += ok ? 1u : 0u;
        // mode 0: the non-targeted attack (hits = correct); 1: the targeted class's untargeted goal (correct, on the mask);
        // 2: targeted (predicted == target, on the mask)
        nt += (mode == 2 ? p == target : ok) && (mode == 0 || in_mask) ? 1u : 0u;
    }
    nc = wave_reduce_add_u32(nc);
    nt = wave_reduce_add_u32(nt);
    if ((tid & 63) == 0) { atomicAdd(&s_cnt[0], nc); atomicAdd(&s_cnt[1], nt); }
    __syncthreads();
    if (tid == 0) {
        const unsigned n_correct = s_cnt[0], n_tgt = s_cnt[1];
        hist_step[0 * G + g] = (float)n_correct;
        hist_step[1 * G + g] = (float)n_tgt;
#pragma unroll
        for (int k = 0; k < 3; ++k) { hist_step[(2 + k) * G + g] = scal[k * G + g]; scal[k * G + g] = 0.0f; }
        bool fire;
        if (mode == 0) fire = (double)n_correct / 4096.0 < 1.0 / 13.0;                       // nontarget.py:87,95
        else {
            const double acc = (double)n_tgt / (double)n_mask[g];                            // target.py:105,113 (0 / 0 = NaN: never fires)
            fire = mode == 2 ? acc > 0.9 : acc < 1.0 / 13.0;                                 // target.py:116-121
        }
        s_exit = fire && active[g] && exit_step[g] < 0 ? 1 : 0;
    }
    __syncthreads();
    if (!s_exit) return;
    if (rows == 1) {                       // (a batch of rows is copied by nu_latch_snapshot_kernel, a full grid)
        const float *src = x0 + base * 9;
        float *dst = out + base * 9;
        for (int i = tid; i < N * 9; i += blockDim.x) dst[(i % 9) * N + i / 9] = src[i];
    }
    if (tid == 0) { exit_step[g] = step; active[g] = 0; }
}

// the snapshot of a group of several batch rows whose exit test has just fired (exit_step[g] == step): x0 -> out, transposed
__global__ void nu_latch_snapshot_kernel(const float *__restrict__ x0, float *__restrict__ out, const int32_t *__restrict__ exit_step,
                                         int step, int rows, int N, size_t total, const float *__restrict__ dconsts)
{
    if (dconsts) step = __float_as_int(dconsts[0]);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t per_group = (size_t)rows * N * 9;
    if (exit_step[t / per_group] != step) return;
    const size_t row = t / ((size_t)N * 9), i = t % ((size_t)N * 9);       // destination element i = c * N + n of this row
    out[t] = x0[row * N * 9 + (i % N) * 9 + i / N];
}

extern "C" int psg_nu_inverse_tanh(const float *x0, int B, int N, float *w_out, psg_stream stream)
{
    PSG_REQUIRE(x0 && w_out && B > 0 && N > 0, "psg_nu_inverse_tanh: bad argument");
    size_t rows = (size_t)B * N;
    hipLaunchKernelGGL(nu_inverse_tanh_kernel, dim3(grid_for(rows * 3)), dim3(256), 0, (hipStream_t)stream, x0, w_out, rows);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_nu_tanh_color(const float *w, const uint8_t *mask, int B, int N, float *x0, psg_stream stream)
{
    PSG_REQUIRE(w && x0 && B > 0 && N > 0, "psg_nu_tanh_color: bad argument");
    size_t rows = (size_t)B * N;
    hipLaunchKernelGGL(nu_tanh_color_kernel, dim3(grid_for(rows * 3)), dim3(256), 0, (hipStream_t)stream, w, mask, x0, N,
                       rows, 0);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// The "_rooms" entry points serve B INDEPENDENT one-room attacks advanced in lockstep (the reference's tar_NU / NU attack
// called once per room, target.py:62-133 at B = 1; pointsecguard_amd/attacks/torchattacks/attacks/nu.py: nu_attack_rooms):
// every room has its own mask row, its own smoothness term and its own loss sums; a room that has left the loop is frozen.
extern "C" int psg_nu_tanh_color_rooms(const float *w, const uint8_t *mask_rooms, int B, int N, float *x0, psg_stream stream)
{
    PSG_REQUIRE(w && x0 && B > 0 && N > 0, "psg_nu_tanh_color_rooms: bad argument");
    size_t rows = (size_t)B * N;
    hipLaunchKernelGGL(nu_tanh_color_kernel, dim3(grid_for(rows * 3)), dim3(256), 0, (hipStream_t)stream, w, mask_rooms, x0, N,
                       rows, 1);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_nu_f_loss_grad(const float *logp, const int32_t *labels, int target, int rows, int n_cls, float kappa,
                                  float tsign, float *dlogp_out, float *f_sum, int32_t *pred_out, psg_stream stream)
{
    PSG_REQUIRE(logp && dlogp_out && rows > 0, "psg_nu_f_loss_grad: bad argument");
    PSG_REQUIRE(n_cls > 1 && n_cls <= MAXC, "psg_nu_f_loss_grad: n_cls=%d out of range", n_cls);
    PSG_REQUIRE(labels || (target >= 0 && target < n_cls), "psg_nu_f_loss_grad: target class %d out of range", target);
    hipLaunchKernelGGL(nu_f_loss_grad_kernel, dim3(psg::ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream, logp,
                       labels, target, rows, n_cls, kappa, tsign, dlogp_out, f_sum, pred_out, 0);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_nu_f_loss_grad_rooms(const float *logp, const int32_t *labels, int target, int B, int N, int n_cls, float kappa,
                                        float tsign, float *dlogp_out, float *f_sum_rooms, int32_t *pred_out, psg_stream stream)
{
    PSG_REQUIRE(logp && dlogp_out && f_sum_rooms && B > 0 && N > 0, "psg_nu_f_loss_grad_rooms: bad argument");
    PSG_REQUIRE(N % 64 == 0, "psg_nu_f_loss_grad_rooms: N=%d must be a multiple of 64 (one wave never spans two rooms)", N);
    PSG_REQUIRE(n_cls > 1 && n_cls <= MAXC, "psg_nu_f_loss_grad_rooms: n_cls=%d out of range", n_cls);
    PSG_REQUIRE(labels || (target >= 0 && target < n_cls), "psg_nu_f_loss_grad_rooms: target class %d out of range", target);
    const int rows = B * N;
    hipLaunchKernelGGL(nu_f_loss_grad_kernel, dim3(psg::ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream, logp,
                       labels, target, rows, n_cls, kappa, tsign, dlogp_out, f_sum_rooms, pred_out, N);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_gcn_f_loss_grad(const float *logits, const int32_t *labels, int target, const uint8_t *mask, int mode,
                                   int rows, int n_point, int n_cls, float kappa, float tsign, float scale,
                                   float *dlogits_out, float *f_sum, int32_t *pred_out, psg_stream stream)
{
    PSG_REQUIRE(logits && dlogits_out && rows > 0 && n_point > 0, "psg_gcn_f_loss_grad: bad argument");
    PSG_REQUIRE(mode >= 0 && mode <= 2, "psg_gcn_f_loss_grad: mode %d out of range", mode);
    PSG_REQUIRE(n_cls > 1 && n_cls <= MAXC, "psg_gcn_f_loss_grad: n_cls=%d out of range", n_cls);
    PSG_REQUIRE((mode != 2 && labels) || (target >= 0 && target < n_cls), "psg_gcn_f_loss_grad: labels / target missing");
    hipLaunchKernelGGL(gcn_f_loss_grad_kernel, dim3(psg::ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                       labels, target, mask, mode, rows, n_point, n_cls, kappa, tsign, scale, dlogits_out, f_sum, pred_out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

static int smooth_knn_launch(const float *adv_color, int adv_stride, size_t adv_room_stride, const float *ref_color, int ref_stride,
                             size_t ref_room_stride, int B, int N, int nb, float *dist_sum, float *grad_out, psg_stream stream,
                             int32_t *nn_io = nullptr, int have_prev = 0)
{
    PSG_REQUIRE(adv_color && ref_color && grad_out && N > 0 && B > 0 && B <= 65535, "psg_smooth_knn: bad argument");
    PSG_REQUIRE(nb > 0 && nb <= SM_MAX_NB, "psg_smooth_knn: neighbour count %d out of range (1..%d)", nb, SM_MAX_NB);
    PSG_REQUIRE(N <= 8192, "psg_smooth_knn: N=%d exceeds the LDS-resident limit 8192", N);
    // ref_color == adv_color selects the ResGCN variants' smooth(adv, adv): gradient through both arguments
    const int symmetric = ref_color == adv_color && ref_stride == adv_stride;
    if (symmetric) PSG_CHECK_HIP(hipMemsetAsync(grad_out, 0, (size_t)B * N * 3 * sizeof(float), (hipStream_t)stream));
    // list length per lane: the neighbour count itself for the usual nb = 5 (every step of the insertion network counts:
    // with 64 lanes a wave enters it in most passes), else 8 or 16
    const bool wide = (size_t)B * psg::ceil_div(N, SM_T / 4) >= 256;   // enough 256-query workgroups to fill the chip
    const int sub = wide ? 4 : 16;
    const size_t lds = (size_t)4 * sub * ((((N + sub - 1) / sub + 3) & ~3) + 8) * sizeof(float);
    const dim3 grid(psg::ceil_div(N, SM_T / (wide ? 4 : 16)), B);
#define PSG_SMOOTH_LAUNCH(NBT, SUB)                                                                                              \
    do {                                                                                                                         \
        if (lds > 48 * 1024)                                                                                                     \
            PSG_CHECK_HIP(hipFuncSetAttribute((const void *)smooth_knn_kernel<NBT, SUB>,                                         \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
        hipLaunchKernelGGL((smooth_knn_kernel<NBT, SUB>), grid, dim3(SM_T), lds, (hipStream_t)stream, adv_color, adv_stride,     \
                           ref_color, ref_stride, N, nb, dist_sum, grad_out, symmetric, adv_room_stride, ref_room_stride,        \
                           nn_io, have_prev);                                                                                    \
    } while (0)
    if (wide) {
        if (nb <= 5) PSG_SMOOTH_LAUNCH(5, 4);
        else if (nb <= 8) PSG_SMOOTH_LAUNCH(8, 4);
        else PSG_SMOOTH_LAUNCH(16, 4);
    } else {
        if (nb <= 5) PSG_SMOOTH_LAUNCH(5, 16);
        else if (nb <= 8) PSG_SMOOTH_LAUNCH(8, 16);
        else PSG_SMOOTH_LAUNCH(16, 16);
    }
#undef PSG_SMOOTH_LAUNCH
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_smooth_knn(const float *adv_color, int adv_stride, const float *ref_color, int ref_stride, int N,
                              int nb, float *dist_sum, float *grad_out, psg_stream stream)
{
    return smooth_knn_launch(adv_color, adv_stride, 0, ref_color, ref_stride, 0, 1, N, nb, dist_sum, grad_out, stream);
}

// B rooms at once: room b's colours start at adv_color + b * adv_room_stride floats (likewise ref); dist_sum [B], grad_out [B][N][3]
extern "C" int psg_smooth_knn_rooms(const float *adv_color, int adv_stride, size_t adv_room_stride, const float *ref_color,
                                    int ref_stride, size_t ref_room_stride, int B, int N, int nb, float *dist_sum_rooms,
                                    float *grad_out, int32_t *nn_state, int have_prev, psg_stream stream)
{
    PSG_REQUIRE(adv_color != ref_color, "psg_smooth_knn_rooms: the symmetric variant is one room at a time");
    return smooth_knn_launch(adv_color, adv_stride, adv_room_stride, ref_color, ref_stride, ref_room_stride, B, N, nb,
                             dist_sum_rooms, grad_out, stream, nn_state, nn_state ? have_prev : 0);
}

extern "C" int psg_nu_adam_step(float *w, float *m, float *v, const uint8_t *mask, const float *dx0, const float *x0,
                                const float *ori, const float *smooth_grad, float c_smooth, float c_l2, float lr,
                                float beta1, float beta2, float eps, int step, int B, int N, float *l2_sum,
                                psg_stream stream)
{
    PSG_REQUIRE(w && m && v && dx0 && x0 && ori && B > 0 && N > 0 && step >= 1, "psg_nu_adam_step: bad argument");
    // torch.optim.Adam (single tensor): step_size = lr / (1 - beta1^t), denom = sqrt(v)/sqrt(1 - beta2^t) + eps
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    size_t rows = (size_t)B * N;
    hipLaunchKernelGGL(nu_adam_step_kernel, dim3(grid_for(rows * 3)), dim3(256), 0, (hipStream_t)stream, w, m, v, mask,
                       dx0, x0, ori, smooth_grad, c_smooth, c_l2, beta1, beta2, eps, step_size, bc2_sqrt, N, rows, l2_sum,
                       (const uint8_t *)nullptr, g_nu_dconsts);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// the same update for B one-room attacks in lockstep: mask [B][N], smooth_grad [B][N][3], l2_sum [B]; room_active [B] (or NULL)
extern "C" int psg_nu_adam_step_rooms(float *w, float *m, float *v, const uint8_t *mask_rooms, const float *dx0, const float *x0,
                                      const float *ori, const float *smooth_grad_rooms, float c_smooth, float c_l2, float lr,
                                      float beta1, float beta2, float eps, int step, int B, int N, const uint8_t *room_active,
                                      float *l2_sum_rooms, psg_stream stream)
{
    PSG_REQUIRE(w && m && v && dx0 && x0 && ori && B > 1 && B <= 65535 && N > 0 && step >= 1, "psg_nu_adam_step_rooms: bad argument (B >= 2)");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    // (12 workgroups per room, four elements per thread at 4096 points: 12 atomics per room sum)
    hipLaunchKernelGGL(nu_adam_step_kernel, dim3(std::min(grid_for((size_t)N * 3), 12), B), dim3(256), 0, (hipStream_t)stream, w, m, v, mask_rooms,
                       dx0, x0, ori, smooth_grad_rooms, c_smooth, c_l2, beta1, beta2, eps, step_size, bc2_sqrt, N, (size_t)N, l2_sum_rooms,
                       room_active, g_nu_dconsts);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// Per-step statistics + exit latch of the NU attacks (see nu_step_latch_kernel): G attacks of `rows` batch rows each.
// pred / labels [G*rows][N]; mask [G][N] or NULL (mode 0); n_mask [G] (modes 1, 2); scal [3][G] = this step's f, Smooth, L2
// sums (zeroed by the call); hist_step [5][G] receives n_correct, n_hits, f, Smooth, L2; x0 [G*rows][N][9], out
// [G*rows][9][N]; active [G] bytes, exit_step [G] (-1 until the group's exit test fires).
extern "C" int psg_nu_step_latch(const int32_t *pred, const int32_t *labels, int target, const uint8_t *mask_groups,
                                 const int32_t *n_mask, int G, int rows, int N, int mode, float *scal, float *hist_step,
                                 const float *x0, float *out_cn, uint8_t *active, int32_t *exit_step, int step, psg_stream stream)
{
    PSG_REQUIRE(pred && labels && scal && hist_step && x0 && out_cn && active && exit_step && G > 0 && rows > 0 && N > 0,
                "psg_nu_step_latch: null / empty argument");
    PSG_REQUIRE(mode == 0 || (mode >= 1 && mode <= 2 && mask_groups && n_mask), "psg_nu_step_latch: modes 1 and 2 need mask and n_mask");
    hipLaunchKernelGGL(nu_step_latch_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, pred, labels, target, mask_groups, n_mask, rows,
                       N, mode, scal, hist_step, x0, out_cn, active, exit_step, step, G, g_nu_dconsts);
    PSG_LAUNCH_CHECK();
    if (rows > 1) {
        const size_t total = (size_t)G * rows * N * 9;
        hipLaunchKernelGGL(nu_latch_snapshot_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0, out_cn,
                           exit_step, step, rows, N, total, g_nu_dconsts);
        PSG_LAUNCH_CHECK();
    }
    return PSG_OK;
}

// One step sequence of a window (shared by the eager and the captured path); dconsts: the step's device row or null.
static int nu_window_steps(const psg_nu_window_args *a, const float *dconsts_rows, psg_stream stream)
{
    const bool rooms = a->G > 1;
    const int B = a->G * a->rows, N = a->N, G = a->G;
    int rc = PSG_OK;
    for (int i = 0; i < a->n_steps && rc == PSG_OK; ++i) {
        const int step = a->step0 + i, slot = a->slot0 + i, adam_t = a->adam_t0 + i + 1;
        const int32_t *f_labels = a->use_target ? nullptr : a->labels;
        const int f_target = a->use_target ? a->target : 0;
        if ((rc = rooms ? psg_nu_tanh_color_rooms(a->w, a->mask, B, N, a->x0, stream) : psg_nu_tanh_color(a->w, a->mask, B, N, a->x0, stream)))
            break;
        if ((rc = psg::pn2_forward_lean(a->model, a->ws, slot, a->x0, a->logp, stream))) break;
        if ((rc = rooms ? psg_nu_f_loss_grad_rooms(a->logp, f_labels, f_target, B, N, PSG_PN2_NUM_CLASSES, a->kappa, a->tsign, a->dlogp,
                                                   a->scal, a->pred, stream)
                        : psg_nu_f_loss_grad(a->logp, f_labels, f_target, B * N, PSG_PN2_NUM_CLASSES, a->kappa, a->tsign, a->dlogp, a->scal,
                                             a->pred, stream)))
            break;
        if ((rc = psg::pn2_backward_colour(a->model, a->ws, slot, a->dlogp, a->dx0, stream))) break;   // (the Adam step reads the colours only)
        if ((rc = psg_smooth_knn_rooms(a->x0 + 3, 9, (size_t)N * 9, a->ori, 3, (size_t)N * 3, G, N, a->neighbour, a->scal + G, a->sgrad,
                                       a->nn_state, (i > 0 || a->warm_first) ? 1 : 0, stream)))
            break;
        g_nu_dconsts = dconsts_rows ? dconsts_rows + 4 * i : nullptr;
        rc = rooms ? psg_nu_adam_step_rooms(a->w, a->m, a->v, a->mask, a->dx0, a->x0, a->ori, a->sgrad, a->c_smooth, a->c_l2, a->lr,
                                            a->beta1, a->beta2, a->eps, adam_t, B, N, a->active, a->scal + 2 * G, stream)
                   : psg_nu_adam_step(a->w, a->m, a->v, a->mask, a->dx0, a->x0, a->ori, a->sgrad, a->c_smooth, a->c_l2, a->lr, a->beta1,
                                      a->beta2, a->eps, adam_t, B, N, a->scal + 2, stream);
        if (rc == PSG_OK)
            rc = psg_nu_step_latch(a->pred, a->labels, f_target, a->mode ? a->mask : nullptr, a->mode ? a->n_mask : nullptr, G, a->rows, N,
                                   a->mode, a->scal, a->hist + (size_t)i * 5 * G, a->x0, a->out, a->active, a->exit_step, step, stream);
        g_nu_dconsts = nullptr;
    }
    g_nu_dconsts = nullptr;
    return rc;
}

// A window replayed as a hipGraph.  A one-room attack step is ~30 short launches, and the host's launch rate - not the GPU
// - bounded the reference's one-call-per-room protocol; windows of the same shape (same buffers, same plan slots) differ only
// in {step number, Adam bias corrections, lr}, which the captured kernels read from a device row written before each
// replay.  The handle remembers ONE window shape: the first window of a shape runs eagerly (it also sets kernel
// attributes outside any capture), the second is captured, later ones are replayed.
struct psg_nu_graph {
    hipGraphExec_t exec = nullptr;
    psg_nu_window_args key;
    uint64_t key_model_gen = 0, key_ws_gen = 0;   // generation numbers of key.model / key.ws: addresses can be re-used (psg_common.h)
    bool have_key = false;
    bool capture_failed = false;    // the capture of this key failed once: its windows stay eager, no retry at every window
    psg::CaptureCounters cap;
    float *dconsts = nullptr;       // device [PSG_NU_GRAPH_MAX_STEPS][4]
    float host[PSG_NU_GRAPH_MAX_STEPS * 4];
};

extern "C" int psg_nu_graph_create(psg_nu_graph **out)
{
    PSG_REQUIRE(out, "psg_nu_graph_create: null out pointer");
    psg_nu_graph *g = new psg_nu_graph();
    if (hipMalloc(&g->dconsts, sizeof(g->host)) != hipSuccess) { delete g; psg::set_error("psg_nu_graph_create: hipMalloc failed"); return PSG_ERR_HIP; }
    *out = g;
    return PSG_OK;
}

extern "C" int psg_nu_graph_destroy(psg_nu_graph *g)
{
    if (!g) return PSG_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    (void)hipFree(g->dconsts);
    delete g;
    return PSG_OK;
}

static bool nu_same_shape(const psg_nu_graph *g, psg_nu_window_args y)
{
    psg_nu_window_args x = g->key;
    x.step0 = y.step0 = 0; x.adam_t0 = y.adam_t0 = 0; x.lr = y.lr = 0.0f;      // what the device row carries
    return memcmp(&x, &y, sizeof(x)) == 0 && g->key_model_gen == psg::pn2_model_generation(y.model) &&
           g->key_ws_gen == psg::pn2_ws_generation(y.ws);
}

// out4 = {captures tried, captures failed, windows replayed as a graph, windows that ran eagerly} of this handle (the two
// eager windows every shape starts with included)
extern "C" int psg_nu_graph_stats(const psg_nu_graph *g, long long *out4)
{
    PSG_REQUIRE(g && out4, "psg_nu_graph_stats: null argument");
    out4[0] = g->cap.tried; out4[1] = g->cap.failed; out4[2] = g->cap.replays; out4[3] = g->cap.eager;
    return PSG_OK;
}

extern "C" int psg_pn2_nu_window(const psg_nu_window_args *a, psg_nu_graph *graph, psg_stream stream)
{
    PSG_REQUIRE(a && a->model && a->ws && a->w && a->m && a->v && a->x0 && a->ori && a->labels && a->logp && a->dlogp && a->dx0 &&
                    a->sgrad && a->pred && a->scal && a->nn_state && a->hist && a->out && a->active && a->exit_step,
                "psg_pn2_nu_window: null argument");
    PSG_REQUIRE(a->n_steps > 0 && a->G > 0 && a->rows > 0 && (a->G == 1 || a->rows == 1), "psg_pn2_nu_window: (G, rows) must be (1, B) or (R, 1)");
    hipStream_t st = (hipStream_t)stream;
    // (PSG_NU_NO_GRAPH=1: eager windows, for counter passes - per-dispatch counter rows need per-dispatch launches)
    static const bool no_graph = psg::env_int("PSG_NU_NO_GRAPH", 0) != 0;
    if (!graph || no_graph || a->n_steps > PSG_NU_GRAPH_MAX_STEPS || psg::trace_sync_enabled()) return nu_window_steps(a, nullptr, stream);
    // the step-dependent constants of this window (torch.optim.Adam: step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t))
    for (int i = 0; i < a->n_steps; ++i) {
        const int t = a->adam_t0 + i + 1;
        const double bc1 = 1.0 - pow((double)a->beta1, (double)t), bc2 = 1.0 - pow((double)a->beta2, (double)t);
        graph->host[4 * i] = __builtin_bit_cast(float, a->step0 + i);
        graph->host[4 * i + 1] = (float)((double)a->lr / bc1);
        graph->host[4 * i + 2] = (float)sqrt(bc2);
        graph->host[4 * i + 3] = 0.0f;
    }
    PSG_CHECK_HIP(hipMemcpyAsync(graph->dconsts, graph->host, (size_t)a->n_steps * 16, hipMemcpyHostToDevice, st));
    const bool same = graph->have_key && nu_same_shape(graph, *a);
    if (same && graph->exec) {
        PSG_CHECK_HIP(hipGraphLaunch(graph->exec, st));
        psg::capture_note(&graph->cap, 0, 0, 1, 0);
        return PSG_OK;
    }
    if (same && !graph->capture_failed) {
        // second window of this shape: capture it, then replay.  A capture that fails (refused on the legacy default stream,
        // or invalidated - DESIGN 5i names what can do that) has executed nothing: the window runs eagerly below and a genuine
        // launch error shows again there; the failure is COUNTED (psg_nu_graph_stats, psg_capture_stats) and this shape is
        // not tried again.
        bool ok = false;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int crc = nu_window_steps(a, graph->dconsts, stream);
            hipGraph_t gr = nullptr;
            const hipError_t e = hipStreamEndCapture(st, &gr);
            ok = crc == PSG_OK && e == hipSuccess && gr && hipGraphInstantiate(&graph->exec, gr, nullptr, nullptr, 0) == hipSuccess;
            if (!ok) graph->exec = nullptr;
            if (gr) (void)hipGraphDestroy(gr);
        }
        (void)hipGetLastError();
        graph->capture_failed = !ok;
        psg::capture_note(&graph->cap, 1, ok ? 0 : 1, 0, 0);
        if (ok) {
            PSG_CHECK_HIP(hipGraphLaunch(graph->exec, st));
            psg::capture_note(&graph->cap, 0, 0, 1, 0);
            return PSG_OK;
        }
    }
    if (!same) {
        if (graph->exec) {              // another shape: forget the old one
            PSG_CHECK_HIP(hipStreamSynchronize(st));
            (void)hipGraphExecDestroy(graph->exec);
            graph->exec = nullptr;
        }
        graph->key = *a;
        graph->key_model_gen = psg::pn2_model_generation(a->model);
        graph->key_ws_gen = psg::pn2_ws_generation(a->ws);
        graph->have_key = true;
        graph->capture_failed = false;
    }
    psg::capture_note(&graph->cap, 0, 0, 0, 1);
    return nu_window_steps(a, graph->dconsts, stream);
}
