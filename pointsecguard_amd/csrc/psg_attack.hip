// Attack-loop arithmetic on gfx950: layout transposes, CE-on-log-probs gradient, the fused
// sign-step / L-inf projection update, and segmentation statistics.  All elementwise / tiny
// reductions: HBM-bound by construction (12-52 bytes per point), coalesced along the point index.
//
// Reference (paths relative to /root/reference):
//   PointNet/attacks/torchattacks/attacks/nontarget.py:26,34-39   NB_attack loss + update
//   PointNet/attacks/torchattacks/attacks/target.py:27,36-43      tar_NB_attack loss + update
//   PointNet/NB_nontarget_test_semseg.py:188-211                  acc / per-class I, U, seen
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

__global__ void ce_logp_grad_kernel(const float *__restrict__ logp, const int32_t *__restrict__ labels, int target,
                                    int rows, int rows_active, int n_cls, float scale, float *__restrict__ dlogp,
                                    float *__restrict__ cost)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float *g = dlogp + (size_t)r * n_cls;
    if (r >= rows_active) {
        for (int c = 0; c < n_cls; ++c) g[c] = 0.0f;
        return;
    }
    const float *lp = logp + (size_t)r * n_cls;
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
