// RandLA-Net inference graph + input (colour) gradient + BIM attack loop on gfx950.
//
// Restates RandLA-Net/RandLANet.py:150-190 (inference) and :323-410 (dilated_res_block, building_block,
// relative_pos_encoding, random_sample, nearest_interpolation, gather_neighbour, att_pooling) of the reference, whose
// 1x1 convolutions are conv + bias -> batch_normalization(eps 1e-6) -> leaky_relu(0.2) (helper_tf_util.py:115-170), and
// the BIM update of ares/ares/attack/bim.py:66-98 with its "colper" loss (:110-116).  One cloud per workspace
// (ConfigS3DIS.val_batch_size = 1, helper_tool.py:52), 5 levels, k = 16.
//
// Every 1x1 convolution is one launch of the fp32-MFMA row GEMM (psg_gemm.cuh) with the BatchNorm affine folded into
// the weights on the host and bias + leaky_relu (+ sign bits for the backward pass) in its epilogue; the per-edge
// attention scores are GEMMs over the [N*16, d] neighbour rows.  Around them: small HBM-bound kernels for the relative
// position encoding, neighbour gather / concat, softmax-over-neighbours pooling and its gradient, random-sample max
// pooling, nearest interpolation, and the transposes of the gathers (float atomics: several edges point at one vertex).
// The xyz branch of the local feature aggregation (relative positions and their two MLPs) does not depend on colour:
// it runs in the forward pass, but the colour attack never differentiates through it.
#include <cstdlib>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "psg_common.h"
#include "psg_gemm.cuh"

using namespace psg;

namespace {

constexpr int RL = 5;            // encoder levels
constexpr int RK = 16;           // neighbours
constexpr int RNCLS = 13;
const int kDout[RL] = {16, 64, 128, 256, 512};
const int kRatio[RL] = {4, 4, 4, 4, 2};
constexpr float kSlope = 0.2f;
#ifndef RL_SMALL_TILE_BELOW
#define RL_SMALL_TILE_BELOW 256
#endif

struct RLayer {
    float *w = nullptr;    // [cout][cin]  BatchNorm scale folded in
    float *wt = nullptr;   // [cin][cout]
    float *b = nullptr;    // [cout] folded bias, or null
    int cin = 0, cout = 0;
};

// Skinny layers (K, M <= 32: everything at level 0 -- 8/16-wide features over 40 960 points or 655 360 edge rows): the
// MFMA row GEMM would spend a 128-column tile on 8-32 useful columns, so these run one row per thread on the vector
// pipe with the weights in LDS (broadcast reads): ~K*M FMAs per row against 4 (K + M) bytes of traffic, i.e.
// bandwidth-bound, which is the floor for these shapes.  Same epilogue semantics as gemm_rows_kernel.
template <int EPI>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(GemmArgs a)
{
    __shared__ float s_w[32 * 32 + 32];
    for (int t = threadIdx.x; t < a.M * a.K; t += 256) s_w[t] = a.w[(size_t)(t / a.K) * a.ld_w + (t % a.K)];
    if (threadIdx.x < 32) s_w[1024 + threadIdx.x] = (a.bias && (int)threadIdx.x < a.M) ? a.bias[threadIdx.x] : 0.0f;
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= a.rows) return;
    float x[32];
    const float *in = a.in + (size_t)row * a.ld_in;
    if (((a.K | a.ld_in) & 3) == 0) {   // 16-byte loads: a wave then reads its 64 rows as whole cache lines
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const float4 v = 4 * k4 < a.K ? *(const float4 *)(in + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
            x[4 * k4] = v.x; x[4 * k4 + 1] = v.y; x[4 * k4 + 2] = v.z; x[4 * k4 + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) x[k] = k < a.K ? in[k] : 0.0f;
    }
    float *out = a.out + (size_t)row * a.ld_out;
    unsigned bits = 0;
    const bool vec_out = ((a.M | a.ld_out) & 3) == 0;
    for (int m0 = 0; m0 < a.M; m0 += 4) {
        float z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + u;
            float acc = 0.0f;
            if (m < a.M) {
#pragma unroll
                for (int k = 0; k < 32; ++k)
                    if (k < a.K) acc += x[k] * s_w[m * a.K + k];
                acc += s_w[1024 + m];
                if (EPI == EPI_LRELU) {
                    const bool pos = acc > 0.0f;
                    bits |= (unsigned)pos << m;
                    acc = pos ? acc : 0.2f * acc;
                }
            }
            z[u] = acc;
        }
        const unsigned pb = a.post_mask ? a.post_mask[row] >> m0 : 0xFu;       // (M <= 32: one mask word per row)
        if (vec_out) {
            float4 *o4 = (float4 *)(out + m0);
            float4 v = make_float4(z[0], z[1], z[2], z[3]);
            if (a.accumulate == 1) { const float4 o = *o4; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            if (!(pb & 1u)) v.x *= a.post_slope;
            if (!(pb & 2u)) v.y *= a.post_slope;
            if (!(pb & 4u)) v.z *= a.post_slope;
            if (!(pb & 8u)) v.w *= a.post_slope;
            *o4 = v;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (m0 + u < a.M) {
                    float vv = a.accumulate == 1 ? out[m0 + u] + z[u] : z[u];
                    if (!((pb >> u) & 1u)) vv *= a.post_slope;
                    out[m0 + u] = vv;
                }
        }
    }
    if (EPI == EPI_LRELU && a.mask_out) a.mask_out[row] = bits;
}

// Round 5: the narrow layers of levels 0 and 1 (8 .. 64 output channels over 80 000 .. 330 000 rows per 8-cloud launch, K <= 128)
// on the matrix cores WITHOUT the LDS-staged tile machinery of gemm_rows_kernel: these launches move 100 - 250 MB for a few
// MFMAs per row, and a 256 x 64 tile whose whole K is one or two 32-wide steps has no loop to pipeline - stage, barrier, 8
// MFMAs, barrier, epilogue, at two waves per SIMD (240 registers): 2.0 TB/s (profiles/r04_randla_bench.json); the row-per-thread
// vector kernel of the <= 32-channel layers: 1.45 TB/s.  Here a wave owns 32 rows: lane (row j, half h) reads its row's K
// floats straight in MFMA operand order (one 16-byte load per 8 channels: k = 8 k8 + 4 h + t), the weights' rows likewise
// from L1 / L2 (<= 32 KB per layer), MT = 1 or 2 accumulator tiles of 32 channels, no LDS, no barrier, ~60 registers (8
// waves per SIMD hide the loads).  Epilogue = gemm_rows_kernel's for the options RandLA-Net uses: bias, leaky ReLU + sign
// bits, accumulate, the stored leaky-ReLU derivative (post_mask).
template <int EPI, int MT>
__global__ __launch_bounds__(256) void direct_gemm_kernel(GemmArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const size_t row = ((size_t)blockIdx.x * 4 + wave) * 32 + j;
    const bool live = row < (size_t)a.rows;
    const float *in = a.in + (live ? row : (size_t)a.rows - 1) * a.ld_in + 4 * h;
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const float *wrow[MT];
    bool wlive[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 32 + j;
        wlive[t] = m < a.M;
        wrow[t] = a.w + (size_t)(wlive[t] ? m : 0) * a.ld_w + 4 * h;
    }
    const int k8n = (a.K + 7) >> 3;
    for (int k8 = 0; k8 < k8n; ++k8) {
        const bool kin = 8 * k8 + 4 * h < a.K;                   // (K is a multiple of 4: the upper half of the last chunk may be absent)
        const float4 av = kin ? *(const float4 *)(in + 8 * k8) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const float4 wv = (kin && wlive[t]) ? *(const float4 *)(wrow[t] + 8 * k8) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc[t] = mfma4<false>(wv, av, acc[t]);
        }
    }
    const int nw = (a.M + 31) >> 5;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int cbase = t * 32;
        if (cbase >= a.M) break;
        const unsigned post_bits = (a.post_mask && live) ? a.post_mask[row * nw + t] : 0xFFFFFFFFu;
        unsigned mbits = 0;
        float4 vv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = cbase + 8 * g + 4 * h;
            const float4 b4 = (a.bias && c < a.M) ? *(const float4 *)(a.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 p4 = (a.pre_add && live && c < a.M) ? *(const float4 *)(a.pre_add + row * a.ld_pre + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float z = acc[t][4 * g + u];
                if (a.bias) z += bb[u];
                if (a.pre_add) z += pp[u];
                if (EPI == EPI_LRELU) {
                    const bool pos = z > 0.0f;
                    mbits |= (unsigned)pos << (8 * g + 4 * h + u);
                    z = pos ? z : 0.2f * z;
                }
                e[u] = z;
            }
            vv[g] = make_float4(e[0], e[1], e[2], e[3]);
        }
        if (EPI == EPI_LRELU && a.mask_out) {
            const unsigned other = __shfl_xor(mbits, 32);
            if (h == 0 && live) a.mask_out[row * nw + t] = mbits | other;
        }
        if (live) {
            float *o = a.out + row * a.ld_out + cbase + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (cbase + 8 * g + 4 * h >= a.M) continue;          // (M is a multiple of 4)
                float4 v = vv[g];
                float4 *dst = (float4 *)(o + 8 * g);
                if (a.accumulate == 1) { const float4 old = *dst; v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
                if (a.post_mask) {
                    const unsigned pb = post_bits >> (8 * g + 4 * h);
                    if (!(pb & 1u)) v.x *= a.post_slope;
                    if (!(pb & 2u)) v.y *= a.post_slope;
                    if (!(pb & 4u)) v.z *= a.post_slope;
                    if (!(pb & 8u)) v.w *= a.post_slope;
                }
                *dst = v;
            }
        }
    }
}

// profile of the launches of the call in progress on this thread (psg_rla_prof_enable), by kernel: tag 0 = 64 x 64
// tiles (gemm_rows_kernel<2,2,.,.,1,1>: the point-sized layers), 1 = 128 x 128 tiles, 2 = 256 x 64 tiles, 3 = the
// row-per-thread kernel of the 8-32 channel layers; each launch with its algorithmic FLOPs and bytes (input rows, output
// rows - twice when accumulated into -, addend rows, weights, mask words).  Tags 4-7 = the attentive-pooling kernels
// (att_pool_split_fwd / _bwd, lfa16_fwd / _bwd) with their algorithmic bytes: every per-edge stream once, every gathered
// table once per point
thread_local EvLog *tl_prof = nullptr;

template <int EPI>
int rl_gemm(const GemmArgs &a, hipStream_t st)
{
    const bool skinny = a.M <= 32 && a.K <= 32 && a.rows >= 4096 && !a.addend && !a.gbias && !a.mask_in && !a.pre_add;
    const bool wide = !skinny && a.M <= 64 && a.rows >= 32768;
    const bool small_tile = !skinny && !wide && (size_t)ceil_div(a.rows, 128) * ceil_div(a.M, 128) < RL_SMALL_TILE_BELOW;
    const double bytes = 4.0 * ((double)a.rows * a.K + (double)a.rows * a.M * (a.accumulate ? 2.0 : 1.0) +
                                (a.addend ? (double)a.rows * a.M : 0.0) + (double)a.K * a.M +
                                ((a.mask_out || a.post_mask) ? (double)a.rows * ((a.M + 31) / 32) : 0.0));
    EvScope prof(tl_prof, skinny ? 3 : (wide ? 2 : (small_tile ? 0 : 1)), 2.0 * a.rows * (double)a.K * a.M, st, bytes);
    // (PSG_RLA_NO_DIRECT=1: the tile / row-per-thread kernels of rounds 1-4, for A/B runs)
    static const bool use_direct = psg::env_int("PSG_RLA_NO_DIRECT", 0) == 0;
    if (use_direct && (skinny || wide) && a.K <= 128 && ((a.K | a.M | a.ld_in | a.ld_w | a.ld_out | a.ld_pre) & 3) == 0 && !a.addend && !a.gbias &&
        !a.mask_in && !a.scale && a.accumulate <= 1) {
        const dim3 grid(ceil_div(a.rows, 128));
        if (a.M <= 32) hipLaunchKernelGGL((direct_gemm_kernel<EPI, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((direct_gemm_kernel<EPI, 2>), grid, dim3(256), 0, st, a);
        PSG_LAUNCH_CHECK();
        return PSG_OK;
    }
    if (skinny) {
        hipLaunchKernelGGL(skinny_gemm_kernel<EPI>, dim3(ceil_div(a.rows, 256)), dim3(256), 0, st, a);
        PSG_LAUNCH_CHECK();
        return PSG_OK;
    }
    if (a.M <= 64 && a.rows >= 32768) {   // 64 output channels (level 1): 256-row x 64-column workgroup tiles, no wasted half tile
        hipLaunchKernelGGL((gemm_rows_kernel<4, 1, EPI, false>), dim3(ceil_div(a.rows, 256), 1), dim3(256), 0, st, a);
        PSG_LAUNCH_CHECK();
        return PSG_OK;
    }
    dim3 grid(ceil_div(a.rows, 128), ceil_div(a.M, 128));
    if ((size_t)grid.x * grid.y < RL_SMALL_TILE_BELOW) {   // few 128-wide tiles: 64 x 64 tiles, one MFMA tile per wave
        dim3 small(ceil_div(a.rows, 64), ceil_div(a.M, 64));
        hipLaunchKernelGGL((gemm_rows_kernel<2, 2, EPI, false, 1, 1>), small, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((gemm_rows_kernel<2, 2, EPI, false>), grid, dim3(256), 0, st, a);
    }
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

GemmArgs rl_args(const float *in, int ld_in, const float *w, int ld_w, float *out, int ld_out, int rows, int K, int M)
{
    GemmArgs a;
    a.in = in; a.w = w; a.bias = nullptr; a.gbias = nullptr; a.scale = nullptr; a.shift = nullptr; a.sq = nullptr;
    a.out = out; a.mask_out = nullptr; a.mask_in = nullptr;
    a.rows = rows; a.K = K; a.M = M; a.ld_in = ld_in; a.ld_w = ld_w; a.ld_out = ld_out; a.group_rows = 1;
    a.accumulate = 0; a.addend = nullptr; a.ld_add = 0;
    return a;
}

// out = [leaky_relu](in . W^T + b), sign bits to `mask` ([rows][ceil(cout/32)] words) when given
int conv_fwd(const RLayer &L, const float *in, int ld_in, float *out, int ld_out, int rows, bool act, uint32_t *mask,
             hipStream_t st)
{
    GemmArgs a = rl_args(in, ld_in, L.w, L.cin, out, ld_out, rows, L.cin, L.cout);
    a.bias = L.b;
    a.mask_out = mask;
    return act ? rl_gemm<EPI_LRELU>(a, st) : rl_gemm<EPI_LINEAR>(a, st);
}

// g[row][c] *= (bit ? 1 : slope): gradient through leaky_relu, in place
// (the memory-bound kernels of this file move 16 bytes per thread: channel counts and leading dimensions are multiples of 4)
__global__ void lrelu_bwd_kernel(float *__restrict__ g, int ld, const uint32_t *__restrict__ mask, int M, size_t total4)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const int m4 = M >> 2;
    const size_t row = t / m4;
    const int c = (int)(t - row * m4) * 4;
    const unsigned bits = (mask[row * ((M + 31) >> 5) + (c >> 5)] >> (c & 31)) & 15u;
    if (bits == 15u) return;
    float4 *p = (float4 *)(g + row * ld + c);
    float4 v = *p;
    if (!(bits & 1u)) v.x *= kSlope;
    if (!(bits & 2u)) v.y *= kSlope;
    if (!(bits & 4u)) v.z *= kSlope;
    if (!(bits & 8u)) v.w *= kSlope;
    *p = v;
}

// din (+)= dz . W   (dz already multiplied by the activation's derivative)
// `lrelu_mask`: when this GEMM is the LAST contributor to din and din is the gradient of a leaky-ReLU output, the sign
// bits of that activation: its derivative is applied to what is stored (after the accumulate), no separate pass
int conv_bwd(const RLayer &L, const float *dz, int ld_dz, float *din, int ld_din, int rows, int accumulate, hipStream_t st,
             const uint32_t *lrelu_mask = nullptr)
{
    GemmArgs a = rl_args(dz, ld_dz, L.wt, L.cout, din, ld_din, rows, L.cout, L.cin);
    a.accumulate = accumulate;
    a.post_mask = lrelu_mask;
    a.post_slope = kSlope;
    return rl_gemm<EPI_LINEAR>(a, st);
}

int lrelu_bwd(float *g, int ld, const uint32_t *mask, int rows, int M, hipStream_t st)
{
    if ((M & 3) || (ld & 3)) { set_error("lrelu_bwd: M=%d, ld=%d must be multiples of 4", M, ld); return PSG_ERR_ARG; }
    const size_t total = (size_t)rows * (M >> 2);
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g, ld, mask, M, total);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// relative_pos_encoding (RandLANet.py:347-353): [dist, xyz_i - xyz_j, xyz_i, xyz_j] per edge
__global__ void relpos_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ neigh, size_t edges, float *__restrict__ out)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= edges) return;
    const size_t i = e / RK;
    const int j = neigh[e];
    const float xi = xyz[3 * i], yi = xyz[3 * i + 1], zi = xyz[3 * i + 2];
    const float xj = xyz[3 * (size_t)j], yj = xyz[3 * (size_t)j + 1], zj = xyz[3 * (size_t)j + 2];
    const float rx = xi - xj, ry = yi - yj, rz = zi - zj;
    float *o = out + e * 10;
    o[0] = sqrtf(rx * rx + ry * ry + rz * rz);
    o[1] = rx; o[2] = ry; o[3] = rz; o[4] = xi; o[5] = yi; o[6] = zi; o[7] = xj; o[8] = yj; o[9] = zj;
}

// cat[e][0:h] = f[neigh[e]][0:h]; cat[e][h:2h] = fxyz[e][0:h]     (gather_neighbour + concat, RandLANet.py:337-339)
__global__ void gather_concat_kernel(const float *__restrict__ f, const int32_t *__restrict__ neigh, const float *__restrict__ fxyz,
                                     int h, size_t total4, float *__restrict__ cat)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const int d4 = h >> 1;                       // float4 per row of cat (d = 2 h channels)
    const size_t e = t / d4;
    const int c = (int)(t - e * d4) * 4;
    ((float4 *)cat)[t] = c < h ? *(const float4 *)(f + (size_t)neigh[e] * h + c) : *(const float4 *)(fxyz + e * h + (c - h));
}

// att_pooling core (RandLANet.py:403-405): scores -> softmax over the 16 neighbours (kept in place of the scores),
// agg[n][c] = sum_k cat[n][k][c] * a[n][k][c].  One thread per (point, channel).
// One thread per (point, 4 channels); every channel's arithmetic is the scalar sequence (max, exp, sum in k order).
__global__ void att_pool_fwd_kernel(const float *__restrict__ cat, float *__restrict__ s, int d, size_t total4, float *__restrict__ agg)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const int d4 = d >> 2;
    const size_t n = t / d4;
    const int c = (int)(t - n * d4) * 4;
    float4 v[RK];
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        v[k] = *(const float4 *)(s + (n * RK + k) * d + c);
        m.x = fmaxf(m.x, v[k].x); m.y = fmaxf(m.y, v[k].y); m.z = fmaxf(m.z, v[k].z); m.w = fmaxf(m.w, v[k].w);
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        v[k].x = expf(v[k].x - m.x); v[k].y = expf(v[k].y - m.y); v[k].z = expf(v[k].z - m.z); v[k].w = expf(v[k].w - m.w);
        sum.x += v[k].x; sum.y += v[k].y; sum.z += v[k].z; sum.w += v[k].w;
    }
    const float4 inv = make_float4(1.0f / sum.x, 1.0f / sum.y, 1.0f / sum.z, 1.0f / sum.w);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const float4 a = make_float4(v[k].x * inv.x, v[k].y * inv.y, v[k].z * inv.z, v[k].w * inv.w);
        *(float4 *)(s + (n * RK + k) * d + c) = a;
        const float4 x = *(const float4 *)(cat + (n * RK + k) * d + c);
        acc.x += x.x * a.x; acc.y += x.y * a.y; acc.z += x.z * a.z; acc.w += x.w * a.w;
    }
    ((float4 *)agg)[t] = acc;
}

// gradient of the pooling: dcat = a * dagg (direct path); ds = a * (g - sum_j a_j g_j), g = cat * dagg (softmax path)
__global__ void att_pool_bwd_kernel(const float *__restrict__ cat, const float *__restrict__ a, const float *__restrict__ dagg, int d,
                                    size_t total2, float *__restrict__ dcat, float *__restrict__ ds)
{
    // one thread per (point, 2 channels): 8-byte accesses, the 16 attention weights and products of both channels in registers
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total2) return;
    const int d2 = d >> 1;
    const size_t n = t / d2;
    const int c = (int)(t - n * d2) * 2;
    const float2 g0 = ((const float2 *)dagg)[t];
    float2 av[RK], gv[RK];
    float2 dot = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        av[k] = *(const float2 *)(a + (n * RK + k) * d + c);
        const float2 x = *(const float2 *)(cat + (n * RK + k) * d + c);
        gv[k] = make_float2(x.x * g0.x, x.y * g0.y);
        dot.x += av[k].x * gv[k].x; dot.y += av[k].y * gv[k].y;
    }
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        *(float2 *)(dcat + (n * RK + k) * d + c) = make_float2(av[k].x * g0.x, av[k].y * g0.y);
        *(float2 *)(ds + (n * RK + k) * d + c) = make_float2(av[k].x * (gv[k].x - dot.x), av[k].y * (gv[k].y - dot.y));
    }
}


// ---- Level 0 (d = 16 channels, h = 8): one attentive pooling in ONE kernel, nothing per-edge written.
// The unfused chain moves the [N*16, 16] edge tensor seven times forward (gather + concat out, score GEMM in / out,
// softmax pooling in / in / out) and about eight times backward; at 16 channels the score "GEMM" is 256 FMAs per edge, which
// a thread does in registers.  Forward: thread (point, channel pair) holds the 2 rows of the score weights it needs,
// walks the 16 neighbours (gathered feature row + position-encoding row = the 16 inputs), keeps its 16 x 2 scores, then
// softmax over the neighbours and the weighted sum, in the arithmetic order of skinny_gemm_kernel / att_pool_fwd_kernel.
// Backward recomputes scores and attention weights instead of reading them back (RandLANet.py:398-410).
__device__ __forceinline__ void lfa16_load_x(const float *__restrict__ f, const float *__restrict__ fxyz, int j, size_t e, float (&x)[16])
{
    const float4 a0 = *(const float4 *)(f + (size_t)j * 8), a1 = *(const float4 *)(f + (size_t)j * 8 + 4);
    const float4 b0 = *(const float4 *)(fxyz + e * 8), b1 = *(const float4 *)(fxyz + e * 8 + 4);
    x[0] = a0.x; x[1] = a0.y; x[2] = a0.z; x[3] = a0.w; x[4] = a1.x; x[5] = a1.y; x[6] = a1.z; x[7] = a1.w;
    x[8] = b0.x; x[9] = b0.y; x[10] = b0.z; x[11] = b0.w; x[12] = b1.x; x[13] = b1.y; x[14] = b1.z; x[15] = b1.w;
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// the thread's own 2 channels of cat[e]: features for q < 4, position encoding for q >= 4
__device__ __forceinline__ float2 lfa16_load_own(const float *__restrict__ f, const float *__restrict__ fxyz, int j, size_t e, int q)
{
    return q < 4 ? *(const float2 *)(f + (size_t)j * 8 + 2 * q) : *(const float2 *)(fxyz + e * 8 + 2 * (q - 4));
}
// Round 5: the 16 x 16 inputs of a point's pooling - cat[e] = [f[neigh[e]] (8) | fxyz[e] (8)] for its 16 edges - are staged in
// LDS once per workgroup (32 points), each edge row fetched by ONE thread.  Rounds 2-4 let each of the point's 8 threads
// (a channel pair each) fetch all 16 rows itself: 80 global 16- / 8-byte loads per thread, 3 GB through the CUs' vector
// memory path per launch of 8 clouds - the kernels ran at 1.2 TB/s of HBM traffic and 10 % of the vector peak, bound by
// neither but by that path and its latency.  Same FMAs in the same order: bit-identical results.
constexpr int LFA_PTS = 32;                       // points per 256-thread workgroup
constexpr int LFA_PSTRIDE = RK * 16 + 4;          // floats per staged point: + 4 de-phases the 8 points a wave reads at once over the banks
__device__ __forceinline__ void lfa16_stage(const float *__restrict__ f, const int32_t *__restrict__ neigh, const float *__restrict__ fxyz,
                                            size_t p0, size_t n, float *__restrict__ s_x, int32_t *__restrict__ s_nb)
{
    for (int e = threadIdx.x; e < LFA_PTS * RK; e += 256) {
        const size_t p = p0 + (size_t)(e >> 4);
        const size_t pe = (p < n ? p : n - 1) * RK + (e & 15);
        const int j = neigh[pe];
        const float4 a0 = *(const float4 *)(f + (size_t)j * 8), a1 = *(const float4 *)(f + (size_t)j * 8 + 4);
        const float4 b0 = *(const float4 *)(fxyz + pe * 8), b1 = *(const float4 *)(fxyz + pe * 8 + 4);
        float *d = s_x + (e >> 4) * LFA_PSTRIDE + (e & 15) * 16;
        *(float4 *)d = a0; *(float4 *)(d + 4) = a1; *(float4 *)(d + 8) = b0; *(float4 *)(d + 12) = b1;
        if (s_nb) s_nb[e] = j;
    }
    __syncthreads();
}

// scores of the 16 neighbours for the thread's 2 channels, and their softmax over the neighbours (in place); xs = the point's
// staged rows
__device__ __forceinline__ void lfa16_attention(const float *__restrict__ xs, const float (&wr)[2][16], const float (&bias)[2], float (&a)[RK][2])
{
    float m[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        float x[16];
        {
            const float4 *r = (const float4 *)(xs + k * 16);
            const float4 a0 = r[0], a1 = r[1], b0 = r[2], b1 = r[3];
            x[0] = a0.x; x[1] = a0.y; x[2] = a0.z; x[3] = a0.w; x[4] = a1.x; x[5] = a1.y; x[6] = a1.z; x[7] = a1.w;
            x[8] = b0.x; x[9] = b0.y; x[10] = b0.z; x[11] = b0.w; x[12] = b1.x; x[13] = b1.y; x[14] = b1.z; x[15] = b1.w;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += x[i] * wr[c][i];
            acc += bias[c];
            a[k][c] = acc;
            m[c] = fmaxf(m[c], acc);
        }
    }
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < RK; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) { a[k][c] = expf(a[k][c] - m[c]); sum[c] += a[k][c]; }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float inv = 1.0f / sum[c];
#pragma unroll
        for (int k = 0; k < RK; ++k) a[k][c] *= inv;
    }
}

// thread = (point, channel pair): 8 threads per point (4 channels per thread took 216 - 256 registers: 1 - 2 waves per SIMD)
__global__ __launch_bounds__(256, 4) void lfa16_fwd_kernel(const float *__restrict__ f, const int32_t *__restrict__ neigh,
                                                        const float *__restrict__ fxyz, const float *__restrict__ w,
                                                        const float *__restrict__ b, size_t n, float *__restrict__ agg)
{
    __shared__ float s_x[LFA_PTS * LFA_PSTRIDE];
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t p = t >> 3;
    const int q = (int)(t & 7);
    lfa16_stage(f, neigh, fxyz, (size_t)blockIdx.x * LFA_PTS, n, s_x, nullptr);
    if (p >= n) return;
    float wr[2][16], bias[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bias[c] = b ? b[2 * q + c] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) wr[c][i] = w[(2 * q + c) * 16 + i];
    }
    const float *xs = s_x + (threadIdx.x >> 3) * LFA_PSTRIDE;
    float a[RK][2];
    lfa16_attention(xs, wr, bias, a);
    float2 out = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const float2 x = *(const float2 *)(xs + k * 16 + 2 * q);     // the thread's own 2 channels of cat[e]
        out.x += x.x * a[k][0]; out.y += x.y * a[k][1];
    }
    *(float2 *)(agg + p * 16 + 2 * q) = out;
}

// gradient w.r.t. the gathered features: df[neigh[e]][i] += a[e][i] * dagg[i] + sum_c ds[e][c] * W[c][i] for i < 8, with
// ds = a * (g - sum_k a g), g = cat * dagg (the position-encoding half carries no colour gradient).  The sum over the 16
// channels c is split over the 8 threads of the point (each owns 2 rows of W) and closed with three shuffles; lane q then
// adds column q: the 8 lanes of a point hit 32 consecutive bytes of the neighbour's row.
__global__ __launch_bounds__(256, 3) void lfa16_bwd_kernel(const float *__restrict__ f, const int32_t *__restrict__ neigh,
                                                        const float *__restrict__ fxyz, const float *__restrict__ w,
                                                        const float *__restrict__ b, const float *__restrict__ dagg, size_t n,
                                                        float *__restrict__ df, float *__restrict__ dcat8)
{
    __shared__ float s_x[LFA_PTS * LFA_PSTRIDE];
    __shared__ int32_t s_nbs[LFA_PTS * RK];
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = (t >> 3) < n;                 // (all lanes stay for the shuffles)
    const size_t p = live ? (t >> 3) : n - 1;
    const int q = (int)(t & 7);
    lfa16_stage(f, neigh, fxyz, (size_t)blockIdx.x * LFA_PTS, n, s_x, s_nbs);
    float wr[2][16], bias[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bias[c] = b ? b[2 * q + c] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) wr[c][i] = w[(2 * q + c) * 16 + i];
    }
    const float *xs = s_x + (threadIdx.x >> 3) * LFA_PSTRIDE;     // (a dead thread's rows are the clamped point n - 1's: staged the same)
    const int32_t *nb = s_nbs + (threadIdx.x >> 3) * RK;
    float a[RK][2];
    lfa16_attention(xs, wr, bias, a);
    const float2 dg2 = *(const float2 *)(dagg + p * 16 + 2 * q);
    const float dg[2] = {dg2.x, dg2.y};
    float dot[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const float2 x = *(const float2 *)(xs + k * 16 + 2 * q);
        dot[0] += a[k][0] * (x.x * dg[0]); dot[1] += a[k][1] * (x.y * dg[1]);
    }
#pragma unroll 4
    for (int k = 0; k < RK; ++k) {
        const float2 x = *(const float2 *)(xs + k * 16 + 2 * q);
        const float xs[2] = {x.x, x.y};
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float ds = a[k][c] * (xs[c] * dg[c] - dot[c]);
#pragma unroll
            for (int i = 0; i < 8; ++i) part[i] += ds * wr[c][i];
        }
        // the direct path a * dagg belongs to the channel's owner: threads 0 .. 3 own the feature half (columns 2 q, 2 q + 1)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (q < 4 && (i >> 1) == q) part[i] += a[k][i & 1] * dg[i & 1];
        // sum over the point's 8 lanes, result in all of them: two quad permutes and the mirror of the half row (lane i
        // <-> 7 - i, whose quad is already summed), as DPP operands of the adds (a ds_bpermute shuffle per step made this
        // kernel LDS-crossbar bound: 384 per thread)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            part[i] += dpp_f<0xB1>(part[i]);    // quad_perm [1,0,3,2]
            part[i] += dpp_f<0x4E>(part[i]);    // quad_perm [2,3,0,1]
            part[i] += dpp_f<0x141>(part[i]);   // row_half_mirror
        }
        float v = part[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) v = q == i ? part[i] : v;
        if (live) {
            if (dcat8) dcat8[(p * RK + k) * 8 + q] = v;          // per-edge rows, gathered through the inverse list afterwards
            else if (v != 0.0f) atomicAdd(df + (size_t)nb[k] * 8 + q, v);
        }
    }
}

// ---- Levels 1-4: attention scores without a per-edge GEMM.  The score layer is linear in cat = [f[neigh] | fxyz], so
//     s[e] = W . cat[e] = (W1 . f)[neigh[e]] + W2 . fxyz[e] = T[neigh[e]] + S2[e]
// with T = f . W1^T an [n, d] GEMM on POINTS (16x fewer rows than edges) and S2 = fxyz . W2^T a per-edge tensor that depends on
// geometry and weights only: computed once per (cloud, model) with the position encodings and kept where the concatenation used
// to be.  (The same identity as EdgeConv's split in psg_resgcn.hip.)  Forward: one kernel reads T through the neighbour
// index, S2, and the two halves of cat on the fly; nothing but the attention weights (for the backward) and the pooled
// [n, d] is written: the gather + concat pass and the [N*16, d] x [d, d] GEMM are gone.  Backward: ds is summed over the
// in-edges of every point (inverse list) into dT [n, d], and the feature gradient is dT . W1 - again a GEMM on points.
__global__ void att_pool_split_fwd_kernel(const float *__restrict__ f, const int32_t *__restrict__ neigh, const float *__restrict__ fxyz,
                                          const float *__restrict__ T, const float *__restrict__ S2, int h, size_t total2,
                                          float *__restrict__ a_out, float *__restrict__ agg)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (point, 2 channels)
    if (t >= total2) return;
    const int d = 2 * h, d2 = d >> 1;
    const size_t n = t / d2;
    const int c = (int)(t - n * d2) * 2;
    int nb[RK];
#pragma unroll
    for (int k = 0; k < RK; k += 4) {     // (the row of 16 indices as four 16-byte loads)
        const int4 q = *(const int4 *)(neigh + n * RK + k);
        nb[k] = q.x; nb[k + 1] = q.y; nb[k + 2] = q.z; nb[k + 3] = q.w;
    }
    float2 v[RK];
    float2 m = make_float2(-INFINITY, -INFINITY);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const size_t e = n * RK + k;
        const float2 tt = *(const float2 *)(T + (size_t)nb[k] * d + c);
        const float2 s2 = *(const float2 *)(S2 + e * d + c);
        v[k] = make_float2(tt.x + s2.x, tt.y + s2.y);
        m.x = fmaxf(m.x, v[k].x); m.y = fmaxf(m.y, v[k].y);
    }
    float2 sum = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        v[k].x = expf(v[k].x - m.x); v[k].y = expf(v[k].y - m.y);
        sum.x += v[k].x; sum.y += v[k].y;
    }
    const float2 inv = make_float2(1.0f / sum.x, 1.0f / sum.y);
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const size_t e = n * RK + k;
        const float2 a = make_float2(v[k].x * inv.x, v[k].y * inv.y);
        if (a_out) *(float2 *)(a_out + e * d + c) = a;
        const float2 x = c < h ? *(const float2 *)(f + (size_t)nb[k] * h + c) : *(const float2 *)(fxyz + e * h + (c - h));
        acc.x += x.x * a.x; acc.y += x.y * a.y;
    }
    ((float2 *)agg)[t] = acc;
}

// ds[e] = a * (g - sum_k a g), g = cat * dagg, for all d channels; ddir[e] = a * dagg for the feature half (c < h).
// One thread per (point, 2 channels): the 16 attention weights and the 16 products stay in registers, every input is read once
// (4 channels per thread had to read cat twice or drop to one wave per SIMD).
// Round 5: the attention weights are RECOMPUTED here from T[neigh] + S2 - the forward's operations in the forward's order, so
// the same bits - instead of being written per edge by the forward ([E][d] floats, 40 % of its HBM traffic) and read back.
__global__ __launch_bounds__(256) void att_pool_split_bwd_kernel(const float *__restrict__ f, const int32_t *__restrict__ neigh, const float *__restrict__ fxyz,
                                          const float *__restrict__ T, const float *__restrict__ S2, const float *__restrict__ dagg, int h,
                                          size_t total, float *__restrict__ ds, float *__restrict__ ddir)
{
    // one thread per (point, channel): half the registers of the channel-pair version (116 -> about 70: seven waves per SIMD
    // instead of four for a kernel that waits on gathered rows), 256-byte instead of 512-byte wave accesses
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int d = 2 * h;
    const size_t n = t / d;
    const int c = (int)(t - n * d);
    const float g0 = dagg[t];
    float av[RK], gv[RK];
    int nb[RK];
#pragma unroll
    for (int k = 0; k < RK; k += 4) {     // (the row of 16 indices as four 16-byte loads)
        const int4 q = *(const int4 *)(neigh + n * RK + k);
        nb[k] = q.x; nb[k + 1] = q.y; nb[k + 2] = q.z; nb[k + 3] = q.w;
    }
    {
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            av[k] = T[(size_t)nb[k] * d + c] + S2[(n * RK + k) * d + c];
            m = fmaxf(m, av[k]);
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            av[k] = expf(av[k] - m);
            sum += av[k];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int k = 0; k < RK; ++k) av[k] *= inv;
    }
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const size_t e = n * RK + k;
        const float x = c < h ? f[(size_t)nb[k] * h + c] : fxyz[e * h + (c - h)];
        gv[k] = x * g0;
        dot += av[k] * gv[k];
    }
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const size_t e = n * RK + k;
        ds[e * d + c] = av[k] * (gv[k] - dot);
        if (c < h) ddir[e * h + c] = av[k] * g0;
    }
}

// ---- inverse lists (transposes of neigh / up) and the gathers that replace the scatter kernels below
__global__ void iota_kernel(int32_t *__restrict__ v, size_t n)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) v[t] = (int32_t)t;
}
// keys[i] = neigh[(b * nc + p) * 16 + k] for the sampled edge i = (b * nc_sub + p) * 16 + k
__global__ void sampled_keys_kernel(const int32_t *__restrict__ neigh, int nc_sub, int nc, size_t total, int32_t *__restrict__ keys)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t r = i >> 4;
    keys[i] = neigh[((r / nc_sub) * nc + r % nc_sub) * RK + (i & 15)];
}
// off[t] = first position of a key >= t in the ascending array keys[0 .. m), t = 0 .. n_targets
__global__ void lower_bound_kernel(const int32_t *__restrict__ keys, int m, int n_targets, int32_t *__restrict__ off)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_targets) return;
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < t) lo = mid + 1; else hi = mid;
    }
    off[t] = lo;
}
// df[j][c] += sum over the edges e that point at row j (ascending e) of src[e][c], c < h   (transpose of gather_neighbour)
// `mask` (or null): sign bits [rows][ceil(h/32)] of the leaky ReLU whose output gradient df is; this kernel is then the last
// contributor to df and applies the activation's derivative to what it stores
template <bool ASSIGN = false>
__global__ void gather_inv_kernel(const float *__restrict__ src, int ld, int h, const int32_t *__restrict__ off,
                                  const int32_t *__restrict__ ent, size_t total, float *__restrict__ df,
                                  const uint32_t *__restrict__ mask = nullptr)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t j = t / h;
    const int c = (int)(t - j * h);
    float acc = 0.0f;
    const int e1 = off[j + 1];
    int i = off[j];
    for (; i + 4 <= e1; i += 4) {   // four independent row loads in flight, added in list order
        const int e0 = ent[i], ea = ent[i + 1], eb = ent[i + 2], ec = ent[i + 3];
        const float v0 = src[(size_t)e0 * ld + c], v1 = src[(size_t)ea * ld + c], v2 = src[(size_t)eb * ld + c], v3 = src[(size_t)ec * ld + c];
        acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; i < e1; ++i) acc += src[(size_t)ent[i] * ld + c];
    float v = ASSIGN ? acc : df[t] + acc;
    if (mask && !((mask[j * ((h + 31) >> 5) + (c >> 5)] >> (c & 31)) & 1u)) v *= kSlope;
    df[t] = v;
}
// transpose of random_sample's max over neighbours: row j of the level collects, over the sampled edges i = r * 16 + k that
// point at it (ascending), dout[r][c] where arg[r][c] == k
__global__ void pool_max_inv_kernel(const float *__restrict__ dout, const uint8_t *__restrict__ arg, int C, const int32_t *__restrict__ off,
                                    const int32_t *__restrict__ ent, size_t total, float *__restrict__ df,
                                    const uint32_t *__restrict__ mask = nullptr)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t j = t / C;
    const int c = (int)(t - j * C);
    float acc = 0.0f;
    const int e1 = off[j + 1];
    int i = off[j];
    for (; i + 4 <= e1; i += 4) {   // four independent (arg, gradient) pairs in flight, added in list order
        const int e0 = ent[i], ea = ent[i + 1], eb = ent[i + 2], ec = ent[i + 3];
        const size_t r0 = (size_t)(e0 >> 4) * C + c, r1 = (size_t)(ea >> 4) * C + c, r2 = (size_t)(eb >> 4) * C + c, r3 = (size_t)(ec >> 4) * C + c;
        const int a0 = arg[r0], a1 = arg[r1], a2 = arg[r2], a3 = arg[r3];
        const float g0 = dout[r0], g1 = dout[r1], g2 = dout[r2], g3 = dout[r3];
        if (a0 == (e0 & 15)) acc += g0;
        if (a1 == (ea & 15)) acc += g1;
        if (a2 == (eb & 15)) acc += g2;
        if (a3 == (ec & 15)) acc += g3;
    }
    for (; i < e1; ++i) {
        const int e = ent[i];
        const size_t r = (size_t)(e >> 4);
        if (arg[r * C + c] == (e & 15)) acc += dout[r * C + c];
    }
    float v = df[t] + acc;
    if (mask && !((mask[j * ((C + 31) >> 5) + (c >> 5)] >> (c & 31)) & 1u)) v *= kSlope;     // (last contributor: see gather_inv_kernel)
    df[t] = v;
}
// The same with four channels per thread (C a multiple of 4: every level): the arg bytes of an entry come as one 32-bit load
// and the gradients as one 16-byte load instead of four 1-byte and four 4-byte ones; sums per channel in the same list order.
__global__ void pool_max_inv4_kernel(const float *__restrict__ dout, const uint8_t *__restrict__ arg, int C, const int32_t *__restrict__ off,
                                     const int32_t *__restrict__ ent, size_t total4, float *__restrict__ df,
                                     const uint32_t *__restrict__ mask = nullptr)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const int C4 = C >> 2;
    const size_t j = t / C4;
    const int c = (int)(t - j * C4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int e1 = off[j + 1];
    int i = off[j];
    auto add = [&](int e, unsigned a, const float4 &g) {
        const unsigned k = (unsigned)(e & 15);
        if ((a & 0xFFu) == k) acc.x += g.x;
        if (((a >> 8) & 0xFFu) == k) acc.y += g.y;
        if (((a >> 16) & 0xFFu) == k) acc.z += g.z;
        if ((a >> 24) == k) acc.w += g.w;
    };
    for (; i + 4 <= e1; i += 4) {   // four independent (arg, gradient) pairs in flight, added in list order
        const int e0 = ent[i], ea = ent[i + 1], eb = ent[i + 2], ec = ent[i + 3];
        const size_t r0 = (size_t)(e0 >> 4) * C + c, r1 = (size_t)(ea >> 4) * C + c, r2 = (size_t)(eb >> 4) * C + c, r3 = (size_t)(ec >> 4) * C + c;
        const unsigned a0 = *(const unsigned *)(arg + r0), a1 = *(const unsigned *)(arg + r1), a2 = *(const unsigned *)(arg + r2),
                       a3 = *(const unsigned *)(arg + r3);
        const float4 g0 = *(const float4 *)(dout + r0), g1 = *(const float4 *)(dout + r1), g2 = *(const float4 *)(dout + r2),
                     g3 = *(const float4 *)(dout + r3);
        add(e0, a0, g0); add(ea, a1, g1); add(eb, a2, g2); add(ec, a3, g3);
    }
    for (; i < e1; ++i) {
        const int e = ent[i];
        const size_t r = (size_t)(e >> 4) * C + c;
        add(e, *(const unsigned *)(arg + r), *(const float4 *)(dout + r));
    }
    float4 v = *(const float4 *)(df + j * C + c);
    v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += acc.w;
    if (mask) {     // (last contributor: see gather_inv_kernel)
        const unsigned m = mask[j * ((C + 31) >> 5) + (c >> 5)] >> (c & 31);
        if (!(m & 1u)) v.x *= kSlope;
        if (!(m & 2u)) v.y *= kSlope;
        if (!(m & 4u)) v.z *= kSlope;
        if (!(m & 8u)) v.w *= kSlope;
    }
    *(float4 *)(df + j * C + c) = v;
}
// transpose of nearest interpolation: dcoarse[t][c] += sum over the rows n with up[n] == t (ascending) of dcat[n][cs + c];
// dskip[n][c] += dcat[n][c] is done by interp_skip_bwd_kernel
__global__ void interp_inv_kernel(const float *__restrict__ dcat, int cs, int cc, const int32_t *__restrict__ off,
                                  const int32_t *__restrict__ ent, size_t total, float *__restrict__ dcoarse,
                                  const uint32_t *__restrict__ mask = nullptr)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t j = t / cc;
    const int c = (int)(t - j * cc);
    float acc = 0.0f;
    const int e1 = off[j + 1];
    for (int i = off[j]; i < e1; ++i) acc += dcat[(size_t)ent[i] * (cs + cc) + cs + c];
    float v = dcoarse[t] + acc;
    if (mask && !((mask[j * ((cc + 31) >> 5) + (c >> 5)] >> (c & 31)) & 1u)) v *= kSlope;    // (last contributor: see gather_inv_kernel)
    dcoarse[t] = v;
}
__global__ void interp_skip_bwd_kernel(const float *__restrict__ dcat, int cs, int cc, size_t total4, float *__restrict__ dskip)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // total4 = rows * cs / 4
    if (t >= total4) return;
    const int s4 = cs >> 2;
    const size_t n = t / s4;
    const int c = (int)(t - n * s4) * 4;
    const float4 g = *(const float4 *)(dcat + n * (cs + cc) + c);
    float4 *o = (float4 *)(dskip + n * cs + c);
    float4 v = *o;
    v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
    *o = v;
}

// transpose of the neighbour gather: df[neigh[e]][c] += dcat[e][c] for c < h (the xyz half carries no colour gradient)
__global__ void gather_bwd_kernel(const float *__restrict__ dcat, const int32_t *__restrict__ neigh, int h, size_t total,
                                  float *__restrict__ df)
{
    // (one element per thread on purpose: the lanes of an atomic instruction then hit consecutive addresses; four
    // atomics per thread, 16 bytes apart across lanes, took 3.7x as long)
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t e = t / h;
    const int c = (int)(t - e * h);
    const float g = dcat[e * 2 * h + c];
    if (g != 0.0f) atomicAdd(df + (size_t)neigh[e] * h + c, g);
}

// enc = leaky_relu(a + b), sign bits out (dilated_res_block, RandLANet.py:330)
__global__ void add_lrelu_kernel(const float *__restrict__ a, const float *__restrict__ b, int M, size_t total, float *__restrict__ out,
                                 uint32_t *__restrict__ mask)
{
    // one wave per 64 consecutive channels of a row: M is a multiple of 32
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < total;
    const float z = live ? a[t] + b[t] : 0.0f;
    const bool pos = z > 0.0f;
    if (live) out[t] = pos ? z : kSlope * z;
    const unsigned long long bal = __ballot(pos && live);
    const int lane = threadIdx.x & 63;
    if (live && (lane & 31) == 0) {
        const size_t row = t / M;
        const int c = (int)(t - row * M);
        mask[row * (M >> 5) + (c >> 5)] = (uint32_t)(bal >> (lane & 32));
    }
}

// random_sample (RandLANet.py:356-371): out[n'][c] = max_k f[pool[n'][k]][c], arg = first winning k
// out[b][p][.] = xyz[b][p][.] for p < nc_sub (3 floats per point)
__global__ void take_prefix_kernel(const float *__restrict__ xyz, int nc, int nc_sub, size_t total, float *__restrict__ out)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t per = (size_t)nc_sub * 3;
    out[t] = xyz[(t / per) * nc * 3 + t % per];
}

// idx[t] += (t / per_cloud) * stride: per-cloud indices -> row numbers of the cloud-major buffers
__global__ void offset_idx_kernel(int32_t *__restrict__ idx, size_t per_cloud, int stride, size_t total)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) idx[t] += (int32_t)((t / per_cloud) * stride);
}

// (sampled row r = cloud b, point p < nc_sub pools over the neighbours of the SAME point in the finer level: row b * nc + p)
__global__ void pool_max_fwd_kernel(const float *__restrict__ f, const int32_t *__restrict__ pool, int C, size_t total,
                                    int nc_sub, int nc, float *__restrict__ out, uint8_t *__restrict__ arg)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // total = sampled rows * C / 4
    if (t >= total) return;
    const int c4 = C >> 2;
    const size_t r = t / c4;
    const int c = (int)(t - r * c4) * 4;
    const size_t n = (r / nc_sub) * nc + r % nc_sub;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 bk = make_uchar4(0, 0, 0, 0);
#pragma unroll 4
    for (int k = 0; k < RK; ++k) {
        const float4 v = *(const float4 *)(f + (size_t)pool[n * RK + k] * C + c);
        if (v.x > best.x) { best.x = v.x; bk.x = (unsigned char)k; }
        if (v.y > best.y) { best.y = v.y; bk.y = (unsigned char)k; }
        if (v.z > best.z) { best.z = v.z; bk.z = (unsigned char)k; }
        if (v.w > best.w) { best.w = v.w; bk.w = (unsigned char)k; }
    }
    ((float4 *)out)[t] = best;
    ((uchar4 *)arg)[t] = bk;
}

__global__ void pool_max_bwd_kernel(const float *__restrict__ dout, const int32_t *__restrict__ pool, const uint8_t *__restrict__ arg,
                                    int C, size_t total, int nc_sub, int nc, float *__restrict__ df)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // (one element per thread: see gather_bwd_kernel)
    if (t >= total) return;
    const size_t r = t / C;
    const int c = (int)(t - r * C);
    const size_t n = (r / nc_sub) * nc + r % nc_sub;
    const float g = dout[t];
    if (g != 0.0f) atomicAdd(df + (size_t)pool[n * RK + arg[t]] * C + c, g);
}

// decoder input: cat[n] = [skip[n] | coarse[up[n]]]   (nearest_interpolation + concat, RandLANet.py:172-174)
__global__ void interp_concat_kernel(const float *__restrict__ skip, int cs, const float *__restrict__ coarse, int cc,
                                     const int32_t *__restrict__ up, size_t total, float *__restrict__ cat)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // total = rows * (cs + cc) / 4
    if (t >= total) return;
    const int w4 = (cs + cc) >> 2;
    const size_t n = t / w4;
    const int c = (int)(t - n * w4) * 4;
    ((float4 *)cat)[t] = c < cs ? *(const float4 *)(skip + n * cs + c) : *(const float4 *)(coarse + (size_t)up[n] * cc + (c - cs));
}

// its transpose: dskip[n] += dcat[n][:cs] (one writer per element), dcoarse[up[n]] += dcat[n][cs:] (atomics)
__global__ void interp_concat_bwd_kernel(const float *__restrict__ dcat, int cs, int cc, const int32_t *__restrict__ up, size_t total,
                                         float *__restrict__ dskip, float *__restrict__ dcoarse)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // (one element per thread: see gather_bwd_kernel)
    if (t >= total) return;
    const int w = cs + cc;
    const size_t n = t / w;
    const int c = (int)(t - n * w);
    const float g = dcat[t];
    if (c < cs) dskip[n * cs + c] += g;
    else if (g != 0.0f) atomicAdd(dcoarse + (size_t)up[n] * cc + (c - cs), g);
}

// "colper" loss of the BIM attack (bim.py:110-116): sum_n max(0, max_k((1 - onehot) * z)_k - z_y); gradient w.r.t. z.
// mask (TBIM / tar_NUattack, bim.py:393-397, tar_NUattack.py:105-110): the per-point loss is multiplied by mask[n] (the
// points of the origin class); `sign` multiplies the gradient (goal 't': grad = -grad, bim.py:350-351).
__global__ void colper_grad_kernel(const float *__restrict__ z, const int32_t *__restrict__ y, const uint8_t *__restrict__ mask,
                                   float sign, int n, float *__restrict__ dz, float *__restrict__ loss)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.0f;
    if (i < n) {
        const float *zi = z + (size_t)i * RNCLS;
        const int yi = y[i];
        float other = 0.0f;          // the masked entry of the true class is 0 and takes part in the max
        int oi = yi;
        for (int k = 0; k < RNCLS; ++k)
            if (k != yi && zi[k] > other) { other = zi[k]; oi = k; }
        const float real = zi[yi];
        const bool on = other - real > 0.0f && (!mask || mask[i]);
        l = on ? other - real : 0.0f;
        for (int k = 0; k < RNCLS; ++k) dz[(size_t)i * RNCLS + k] = 0.0f;
        if (on) {
            dz[(size_t)i * RNCLS + yi] = -sign;
            if (oi != yi) dz[(size_t)i * RNCLS + oi] = sign;
        }
    }
    if (loss) {
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        if ((threadIdx.x & 63) == 0 && l != 0.0f) atomicAdd(loss, l);
    }
}

// BIM update of the colours (bim.py:84-98), goal 'ut' (ascent).  l_inf: clip(adv + alpha*sign(g), x-eps, x+eps);
// l_2: x + clip_by_norm(adv - x + alpha * g/|g|, eps); then clip to [0, 1], PER CLOUD: norms[b] = |g_b|^2,
// norms[B + b] = |delta_b|^2 for the B clouds of a cloud-batch workspace.
// squared norm of the colour part of ONE CLOUD per workgroup (blockIdx.x = cloud): every thread adds its strided elements in
// double, then a fixed tree - one writer, a fixed order: the l_2 update is bit-reproducible (the atomic sum it replaces was
// reproducible to rounding only) and a cloud-batch workspace gets one norm per cloud, as bim.py:84-98 normalises per sample.
__global__ __launch_bounds__(1024) void sq_norm_kernel(const float *__restrict__ a, int ld, int c0, size_t nc, float *__restrict__ out)
{
    __shared__ double s_part[16];
    const float *p = a + (size_t)blockIdx.x * nc * ld;
    double s = 0.0;
    for (size_t t = threadIdx.x; t < nc * 3; t += blockDim.x) {
        const float v = p[(t / 3) * ld + c0 + (t % 3)];
        s += (double)v * (double)v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
        out[blockIdx.x] = (float)tot;
    }
}

__global__ void bim_linf_kernel(float *__restrict__ feat, const float *__restrict__ dfeat, const float *__restrict__ ori, size_t n,
                                float alpha, float eps)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 3) return;
    const size_t i = t / 3;
    const int c = (int)(t % 3);
    const float g = dfeat[i * 6 + 3 + c], x = ori[t];
    const float sg = g > 0.0f ? 1.0f : (g < 0.0f ? -1.0f : 0.0f);
    float v = feat[i * 6 + 3 + c] + alpha * sg;
    v = fminf(fmaxf(v, x - eps), x + eps);
    feat[i * 6 + 3 + c] = fminf(fmaxf(v, 0.0f), 1.0f);
}

__global__ void bim_l2_delta_kernel(const float *__restrict__ feat, const float *__restrict__ dfeat, const float *__restrict__ ori,
                                    size_t n, size_t nc, float alpha, const float *__restrict__ gnorm2, float *__restrict__ delta)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 3) return;
    const size_t i = t / 3;
    const int c = (int)(t % 3);
    const float gn = fmaxf(1e-12f, sqrtf(gnorm2[i / nc]));            // the cloud's own gradient norm
    delta[t] = feat[i * 6 + 3 + c] - ori[t] + alpha * (dfeat[i * 6 + 3 + c] / gn);
}

__global__ void bim_l2_apply_kernel(float *__restrict__ feat, const float *__restrict__ ori, const float *__restrict__ delta, size_t n,
                                    size_t nc, float eps, const float *__restrict__ dnorm2)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 3) return;
    const size_t i = t / 3;
    const int c = (int)(t % 3);
    const float dn = sqrtf(dnorm2[i / nc]);
    const float scale = dn > eps ? eps / dn : 1.0f;
    feat[i * 6 + 3 + c] = fminf(fmaxf(ori[t] + delta[t] * scale, 0.0f), 1.0f);
}

__global__ void copy_cols_kernel(const float *__restrict__ src, int ld_s, int c0, int nc, size_t rows, float *__restrict__ dst, int ld_d, int d0)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * nc) return;
    const size_t r = t / nc;
    const int c = (int)(t % nc);
    dst[r * ld_d + d0 + c] = src[r * ld_s + c0 + c];
}

inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

struct EncLayers { RLayer mlp1, lfa_mlp1, att1_fc, att1_mlp, lfa_mlp2, att2_fc, att2_mlp, mlp2, shortcut; };

}  // namespace

struct psg_rla_model {
    uint64_t gen = psg::next_generation();   // never re-used (psg_common.h): what the replayed BIM iteration is keyed on
    psg_ctx *ctx;
    RLayer fc0, decoder0, dec[RL], fc1, fc2, fc;
    EncLayers enc[RL];
    std::vector<void *> allocs;
};

struct LevelBuf {
    int n = 0, n_sub = 0, d = 0, h = 0, d_in = 0;   // n, n_sub: rows of ALL clouds of the workspace (B * per-cloud counts)
    int nc = 0, nc_sub = 0;                         // per cloud
    float *xyz = nullptr;                           // [B][nc][3]: the level's points, cloud by cloud
    int32_t *neigh = nullptr, *up = nullptr;   // [n][16], [n][1] (row numbers of this level / of the next); pool = a cloud's first nc_sub rows
    // transposes of the two index arrays (the geometry is fixed for a whole attack, so they are built once per cloud): for every
    // row j of this level the edges e = p * 16 + k with neigh[e] == j, ascending (inv_off [n + 1], inv_ent [n * 16]); for every row
    // t of the next level the rows n with up[n] == t, ascending (invu_off [n_sub + 1], invu_ent [n]).  The backward pass GATHERS
    // through them in that fixed order instead of scattering with float atomics: bit-reproducible, and faster.
    int32_t *inv_off = nullptr, *inv_ent = nullptr, *invu_off = nullptr, *invu_ent = nullptr;
    // the same for the edges of the SAMPLED points only (the max-pool reads just those): entries i = r * 16 + k, r a sampled row
    int32_t *invp_off = nullptr, *invp_ent = nullptr;
    float *relpos, *fxyz1, *fxyz2;             // [n*16][10], [n*16][h] x2
    float *fpc, *cat1, *a1, *agg1, *fagg1, *cat2, *a2, *agg2, *fagg2, *m2, *sc, *enc, *samp;
    uint32_t *m_fpc, *m_fagg1, *m_fagg2, *m_enc;
    uint8_t *arg;
    float *d_enc, *d_samp;                     // gradients w.r.t. enc [n][2d] and samp [n_sub][2d]
    float *d_fpc, *d_fagg1;                    // [n][h]
};

struct psg_rla_ws {
    psg_ctx *ctx;
    int N = 0;                      // rows of all clouds: B * Nc
    int B = 1, Nc = 0;              // clouds per workspace (independent: every index stays inside its cloud), points per cloud
    void *arena = nullptr;
    size_t bytes = 0;
    float *xyz_all;                 // [B][Nc][3]; a level's points are the first Nc / ratio points of each cloud
    float *xyz_last;                // [B][nc5][3]: the sub-sampled points of the last level
    LevelBuf lv[RL];
    float *f0; uint32_t *m_f0;      // fc0 output [N][8]
    float *dec0; uint32_t *m_dec0;  // decoder_0 output [n5][1024]
    float *dec_cat[RL], *dec_out[RL]; uint32_t *m_dec[RL];
    float *fc1o, *fc2o; uint32_t *m_fc1, *m_fc2;
    float *logits, *dlogits;
    float *scratch_a, *scratch_b;   // [max edges * d] gradient scratch (dcat / ds), also decoder d_cat
    float *d_f0, *d_dec0, *d_dec_out[RL], *d_fc1o, *d_fc2o;
    char *acc = nullptr; size_t acc_bytes = 0;   // the contiguous block of gradient accumulators
    float *feat, *dfeat, *ori, *delta, *norms;   // attack state: [N][6], [N][6], [N][3], [N][3], [2][B] (l_2: squared gradient / delta norm per cloud)
    int32_t *labels;
    bool cloud_set = false, have_fwd = false;
    bool fuse16 = true;           // PSG_RLA_NO_FUSE16=1: the unfused chain at level 0 too (A/B runs, tests)
    bool split = true;            // PSG_RLA_NO_SPLIT=1: levels 1-4 through the gather + per-edge score GEMM chain (A/B runs)
    bool use_inv = true;          // PSG_RLA_ATOMICS=1: the scatter kernels with float atomics instead of the inverse-list gathers
    size_t scratch_bytes = 0;     // of scratch_a and of scratch_b
    EvLog prof;                   // psg_rla_prof_enable
    uint64_t xyz_branch_model = 0;   // generation number (not the address: a freed model's address can come back) of the model whose xyz-branch features (fxyz1 / fxyz2) are resident
    // hipGraph of one BIM iteration (forward, loss gradient, backward, update: ~150 short launches), valid for the
    // (model, eps, alpha, metric) below; every captured kernel works on workspace buffers, so it is cloud-independent
    hipGraphExec_t bim_exec = nullptr;
    uint64_t bim_model_gen = 0;      // the model's generation number, not its address (psg_common.h)
    float bim_eps = 0.f, bim_alpha = 0.f;
    int bim_metric = -1;
    bool bim_capture_failed = false; // the capture for this key failed once: stay eager instead of trying in every call
    psg::CaptureCounters cap;
};

namespace {

template <typename T> T *upload(psg_rla_model *m, const std::vector<T> &h)
{
    void *p = nullptr;
    if (hipMalloc(&p, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)psg::copy_sync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    m->allocs.push_back(p);
    return (T *)p;
}

// tensors[6 * i .. 6 * i + 5] = weight [cout][cin], bias or null, bn gamma, beta, mean, var or null
RLayer make_layer(psg_rla_model *m, const float *const *t, int cin, int cout)
{
    std::vector<float> w((size_t)cout * cin), wt((size_t)cin * cout), b(cout, 0.0f);
    for (int o = 0; o < cout; ++o) {
        double s = 1.0, sh = 0.0;
        if (t[2]) { s = (double)t[2][o] / sqrt((double)t[5][o] + 1e-6); sh = (double)t[3][o] - (double)t[4][o] * s; }
        for (int k = 0; k < cin; ++k) {
            const float v = (float)((double)t[0][(size_t)o * cin + k] * s);
            w[(size_t)o * cin + k] = v;
            wt[(size_t)k * cout + o] = v;
        }
        b[o] = (float)((t[1] ? (double)t[1][o] : 0.0) * s + sh);
    }
    RLayer L;
    L.cin = cin; L.cout = cout;
    L.w = upload(m, w); L.wt = upload(m, wt);
    L.b = (t[1] || t[2]) ? upload(m, b) : nullptr;
    return L;
}

}  // namespace

extern "C" int psg_rla_model_create(psg_ctx *ctx, const float *const *tensors, int n_tensors, psg_rla_model **out)
{
    PSG_REQUIRE(ctx && tensors && out, "psg_rla_model_create: null argument");
    PSG_REQUIRE(n_tensors == 6 * PSG_RLA_NUM_LAYERS, "psg_rla_model_create: expected %d tensors (6 per layer), got %d",
                6 * PSG_RLA_NUM_LAYERS, n_tensors);
    PSG_CHECK_HIP(hipSetDevice(ctx->device));
    auto *m = new psg_rla_model();
    m->ctx = ctx;
    for (int l = 0; l < PSG_RLA_NUM_LAYERS; ++l)
        if (!tensors[6 * l]) { delete m; set_error("psg_rla_model_create: weight of layer %d is null", l); return PSG_ERR_ARG; }
    int li = 0;
    auto next = [&](int cin, int cout) {
        RLayer L = make_layer(m, tensors + 6 * li, cin, cout);
        ++li;
        return L;
    };
    m->fc0 = next(6, 8);
    int d_in = 8;
    for (int i = 0; i < RL; ++i) {
        const int d = kDout[i], h = d / 2;
        EncLayers &E = m->enc[i];
        E.mlp1 = next(d_in, h); E.lfa_mlp1 = next(10, h); E.att1_fc = next(d, d); E.att1_mlp = next(d, h);
        E.lfa_mlp2 = next(h, h); E.att2_fc = next(d, d); E.att2_mlp = next(d, d); E.mlp2 = next(d, 2 * d);
        E.shortcut = next(d_in, 2 * d);
        d_in = 2 * d;
    }
    m->decoder0 = next(d_in, d_in);
    const int enc_c[RL + 1] = {2 * kDout[0], 2 * kDout[0], 2 * kDout[1], 2 * kDout[2], 2 * kDout[3], 2 * kDout[4]};
    int feat = d_in;
    for (int j = 0; j < RL; ++j) {
        const int skip = enc_c[RL - 1 - j];
        m->dec[j] = next(skip + feat, skip);
        feat = skip;
    }
    m->fc1 = next(feat, 64); m->fc2 = next(64, 32); m->fc = next(32, RNCLS);
    for (void *p : m->allocs)
        if (!p) { set_error("psg_rla_model_create: device allocation failed"); return PSG_ERR_HIP; }
    *out = m;
    return PSG_OK;
}

extern "C" int psg_rla_model_destroy(psg_rla_model *m)
{
    if (!m) return PSG_OK;
    for (void *p : m->allocs) (void)hipFree(p);
    delete m;
    return PSG_OK;
}

extern "C" int psg_rla_ws_create(psg_ctx *ctx, int n_points, psg_rla_ws **out) { return psg_rla_ws_create_batch(ctx, n_points, 1, out); }

// A workspace for `batch` clouds of n_points each, attacked together: rows of every buffer are cloud-major, every index
// (neighbours, pooling, interpolation) stays inside its cloud, so the clouds are as independent as in `batch` workspaces;
// one launch of every kernel serves them all (the per-cloud network is ~300 launches of 5-35 us per iteration).
extern "C" int psg_rla_ws_create_batch(psg_ctx *ctx, int n_points, int batch, psg_rla_ws **out)
{
    PSG_REQUIRE(ctx && out, "psg_rla_ws_create: null argument");
    PSG_REQUIRE(n_points >= 8192 && n_points <= 65536 && n_points % 512 == 0,
                "psg_rla_ws_create: n_points=%d must be a multiple of 512 in [8192, 65536] (five sub-samplings, >= 16 points left)", n_points);
    PSG_REQUIRE(batch >= 1 && (size_t)batch * n_points <= (size_t)1 << 20, "psg_rla_ws_create: batch=%d x n_points=%d exceeds 2^20 rows",
                batch, n_points);
    PSG_CHECK_HIP(hipSetDevice(ctx->device));
    auto *ws = new psg_rla_ws();
    ws->ctx = ctx; ws->B = batch; ws->Nc = n_points; ws->N = batch * n_points;
    { const char *nf = psg::env_str("PSG_RLA_NO_FUSE16"); ws->fuse16 = !(nf && atoi(nf)); }
    { const char *at = psg::env_str("PSG_RLA_ATOMICS"); ws->use_inv = !(at && atoi(at)); }
    { const char *ns = psg::env_str("PSG_RLA_NO_SPLIT"); ws->split = !(ns && atoi(ns)) && ws->use_inv; }
    for (int pass = 0; pass < 2; ++pass) {
        size_t off = 0;
        auto take = [&](size_t bytes) {
            off = (off + 255) & ~(size_t)255;
            char *p = pass ? (char *)ws->arena + off : nullptr;
            off += bytes;
            return (void *)p;
        };
        const size_t N = (size_t)ws->N;
        ws->xyz_all = (float *)take(N * 3 * 4);
        int n = ws->N, nc = n_points, d_in = 8;
        size_t max_edge = 0;
        for (int i = 0; i < RL; ++i) {
            LevelBuf &L = ws->lv[i];
            L.n = n; L.n_sub = n / kRatio[i]; L.d = kDout[i]; L.h = L.d / 2; L.d_in = d_in;
            L.nc = nc; L.nc_sub = nc / kRatio[i];
            L.xyz = i == 0 ? ws->xyz_all : (float *)take((size_t)n * 3 * 4);
            const size_t E = (size_t)n * RK, d = L.d, h = L.h;
            max_edge = std::max(max_edge, E * d);
            L.neigh = (int32_t *)take(E * 4); L.up = (int32_t *)take((size_t)n * 4);
            L.inv_off = (int32_t *)take(((size_t)n + 1) * 4); L.inv_ent = (int32_t *)take(E * 4);
            L.invu_off = (int32_t *)take(((size_t)L.n_sub + 1) * 4); L.invu_ent = (int32_t *)take((size_t)n * 4);
            L.invp_off = (int32_t *)take(((size_t)n + 1) * 4); L.invp_ent = (int32_t *)take((size_t)L.n_sub * RK * 4);
            L.relpos = (float *)take(E * 10 * 4); L.fxyz1 = (float *)take(E * h * 4); L.fxyz2 = (float *)take(E * h * 4);
            L.fpc = (float *)take((size_t)n * h * 4); L.cat1 = (float *)take(E * d * 4); L.a1 = (float *)take(E * d * 4);
            L.agg1 = (float *)take((size_t)n * d * 4); L.fagg1 = (float *)take((size_t)n * h * 4);
            L.cat2 = (float *)take(E * d * 4); L.a2 = (float *)take(E * d * 4); L.agg2 = (float *)take((size_t)n * d * 4);
            L.fagg2 = (float *)take((size_t)n * d * 4); L.m2 = (float *)take((size_t)n * 2 * d * 4);
            L.sc = (float *)take((size_t)n * 2 * d * 4); L.enc = (float *)take((size_t)n * 2 * d * 4);
            L.samp = (float *)take((size_t)L.n_sub * 2 * d * 4);
            L.m_fpc = (uint32_t *)take((size_t)n * ceil_div((int)h, 32) * 4); L.m_fagg1 = (uint32_t *)take((size_t)n * ceil_div((int)h, 32) * 4);
            L.m_fagg2 = (uint32_t *)take((size_t)n * ceil_div((int)d, 32) * 4); L.m_enc = (uint32_t *)take((size_t)n * (2 * d / 32) * 4);
            L.arg = (uint8_t *)take((size_t)L.n_sub * 2 * d);
            n = L.n_sub; nc = L.nc_sub; d_in = 2 * L.d;
        }
        const int n5 = ws->lv[RL - 1].n_sub;
        ws->xyz_last = (float *)take((size_t)n5 * 3 * 4);
        ws->f0 = (float *)take(N * 8 * 4); ws->m_f0 = (uint32_t *)take(N * 4);
        ws->dec0 = (float *)take((size_t)n5 * 1024 * 4); ws->m_dec0 = (uint32_t *)take((size_t)n5 * 32 * 4);
        int feat = 1024;
        for (int j = 0; j < RL; ++j) {
            const LevelBuf &L = ws->lv[RL - 1 - j];        // decoder layer j produces features at level RL-1-j's points
            const int skip = j == RL - 1 ? 2 * kDout[0] : 2 * kDout[RL - 2 - j];
            ws->dec_cat[j] = (float *)take((size_t)L.n * (skip + feat) * 4);
            ws->dec_out[j] = (float *)take((size_t)L.n * skip * 4);
            ws->m_dec[j] = (uint32_t *)take((size_t)L.n * ceil_div(skip, 32) * 4);
            max_edge = std::max(max_edge, (size_t)L.n * (skip + feat));
            feat = skip;
        }
        ws->fc1o = (float *)take(N * 64 * 4); ws->fc2o = (float *)take(N * 32 * 4);
        ws->m_fc1 = (uint32_t *)take(N * 2 * 4); ws->m_fc2 = (uint32_t *)take(N * 4);
        ws->logits = (float *)take(N * RNCLS * 4); ws->dlogits = (float *)take(N * RNCLS * 4);
        ws->scratch_a = (float *)take(max_edge * 4); ws->scratch_b = (float *)take(max_edge * 4);
        ws->scratch_bytes = max_edge * 4;
        ws->d_f0 = (float *)take(N * 8 * 4);
        // gradient accumulators (targets of atomics / of several consumers): one contiguous block, zeroed by ONE memset
        // at the start of a backward pass instead of ~25 small ones spread over it
        off = (off + 255) & ~(size_t)255;
        const size_t acc_begin = off;
        for (int i = 0; i < RL; ++i) {
            LevelBuf &L = ws->lv[i];
            L.d_enc = (float *)take((size_t)L.n * 2 * L.d * 4); L.d_samp = (float *)take((size_t)L.n_sub * 2 * L.d * 4);
            L.d_fpc = (float *)take((size_t)L.n * L.h * 4); L.d_fagg1 = (float *)take((size_t)L.n * L.h * 4);
        }
        ws->d_dec0 = (float *)take((size_t)n5 * 1024 * 4);
        for (int j = 0; j < RL; ++j) {
            const LevelBuf &L = ws->lv[RL - 1 - j];
            const int skip = j == RL - 1 ? 2 * kDout[0] : 2 * kDout[RL - 2 - j];
            ws->d_dec_out[j] = (float *)take((size_t)L.n * skip * 4);
        }
        off = (off + 255) & ~(size_t)255;
        ws->acc = pass ? (char *)ws->arena + acc_begin : nullptr;
        ws->acc_bytes = off - acc_begin;
        ws->d_fc1o = (float *)take(N * 64 * 4); ws->d_fc2o = (float *)take(N * 32 * 4);
        ws->feat = (float *)take(N * 6 * 4); ws->dfeat = (float *)take(N * 6 * 4); ws->ori = (float *)take(N * 3 * 4);
        ws->delta = (float *)take(N * 3 * 4); ws->norms = (float *)take((size_t)(2 * batch > 4 ? 2 * batch : 4) * 4); ws->labels = (int32_t *)take(N * 4);
        if (!pass) {
            ws->bytes = (off + 255) & ~(size_t)255;
            hipError_t e = hipMalloc(&ws->arena, ws->bytes);
            if (e != hipSuccess) {
                set_error("psg_rla_ws_create: hipMalloc(%zu) failed: %s", ws->bytes, hipGetErrorString(e));
                delete ws;
                return PSG_ERR_HIP;
            }
        }
    }
    *out = ws;
    return PSG_OK;
}

extern "C" int psg_rla_ws_destroy(psg_rla_ws *ws)
{
    if (!ws) return PSG_OK;
    if (ws->bim_exec) (void)hipGraphExecDestroy(ws->bim_exec);
    if (ws->arena) (void)hipFree(ws->arena);
    ws->prof.destroy();
    delete ws;
    return PSG_OK;
}

extern "C" size_t psg_rla_ws_bytes(const psg_rla_ws *ws) { return ws ? ws->bytes : 0; }

// per-launch HIP-event profile of the GEMM launches (tag 0); while enabled the BIM loop stays eager (no hipGraph)
extern "C" int psg_rla_prof_enable(psg_rla_ws *ws, int on)
{
    PSG_REQUIRE(ws, "psg_rla_prof_enable: null workspace");
    ws->prof.reset(on != 0);
    return PSG_OK;
}

extern "C" int psg_rla_prof_read(psg_rla_ws *ws, int n_tags, double *total_ms, int *counts, double *flops)
{
    // (one tag = every GEMM launch, as before round 3: the per-kernel split is psg_rla_prof_read_kernels)
    PSG_REQUIRE(ws && total_ms && counts && n_tags >= 1, "psg_rla_prof_read: need room for 1 tag");
    double ms[4], fl[4];
    int cnt[4];
    if (ws->prof.read(4, ms, cnt, fl)) { set_error("psg_rla_prof_read: event query failed"); return PSG_ERR_HIP; }
    total_ms[0] = ms[0] + ms[1] + ms[2] + ms[3];
    counts[0] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    if (flops) flops[0] = fl[0] + fl[1] + fl[2] + fl[3];
    for (int i = 1; i < n_tags; ++i) { total_ms[i] = 0.0; counts[i] = 0; if (flops) flops[i] = 0.0; }
    return PSG_OK;
}

extern "C" int psg_rla_prof_read_kernels(psg_rla_ws *ws, int n_tags, double *total_ms, int *counts, double *flops, double *bytes)
{
    PSG_REQUIRE(ws && total_ms && counts && flops && bytes && n_tags >= 4, "psg_rla_prof_read_kernels: need room for 4 tags");
    if (ws->prof.read(n_tags, total_ms, counts, flops, bytes)) { set_error("psg_rla_prof_read_kernels: event query failed"); return PSG_ERR_HIP; }
    return PSG_OK;
}

namespace {
struct ProfBind {   // routes rl_gemm's scopes to the workspace of the call in progress
    EvLog *prev;
    explicit ProfBind(psg_rla_ws *ws) : prev(tl_prof) { tl_prof = (ws && ws->prof.on) ? &ws->prof : nullptr; }
    ~ProfBind() { tl_prof = prev; }
};
}  // namespace

// ent = the positions 0 .. m-1 of keys[], stably sorted by key (hipcub radix sort: ties keep their ascending position),
// off[t] = where target t's run begins.  Temporaries in the workspace's two scratch buffers (idle while the pyramid is built).
static int build_inverse(psg_rla_ws *ws, const int32_t *keys, size_t m, int n_targets, int32_t *off, int32_t *ent, hipStream_t st)
{
    int32_t *iota = (int32_t *)ws->scratch_a, *keys_sorted = iota + m;
    if (2 * m * sizeof(int32_t) > ws->scratch_bytes) { set_error("build_inverse: %zu keys do not fit the scratch buffer", m); return PSG_ERR_STATE; }
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(m)), dim3(256), 0, st, iota, m);
    PSG_LAUNCH_CHECK();
    int bits = 1;
    while ((1 << bits) < n_targets) ++bits;
    size_t tmp_bytes = 0;
    PSG_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_sorted, iota, ent, (int)m, 0, bits, st));
    if (tmp_bytes > ws->scratch_bytes) { set_error("build_inverse: radix sort needs %zu bytes of scratch", tmp_bytes); return PSG_ERR_STATE; }
    PSG_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs((void *)ws->scratch_b, tmp_bytes, keys, keys_sorted, iota, ent, (int)m, 0, bits, st));
    hipLaunchKernelGGL(lower_bound_kernel, dim3(blocks_for((size_t)n_targets + 1)), dim3(256), 0, st, keys_sorted, (int)m, n_targets, off);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// The index pyramid of the reference's tf_map (main_S3DIS.py:198-207) and the relative position encodings, from the
// cloud in ws->xyz_all.
static int build_pyramid(psg_rla_ws *ws, psg_stream stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc;
    for (int i = 0; i < RL; ++i) {
        LevelBuf &L = ws->lv[i];
        // sub_points = the first nc_sub points of every cloud, compacted into the next level's [B][nc_sub][3]
        float *sub = i + 1 < RL ? ws->lv[i + 1].xyz : ws->xyz_last;
        hipLaunchKernelGGL(take_prefix_kernel, dim3(blocks_for((size_t)L.n_sub * 3)), dim3(256), 0, st, L.xyz, L.nc, L.nc_sub,
                           (size_t)L.n_sub * 3, sub);
        PSG_LAUNCH_CHECK();
        if ((rc = psg_knn_points(ws->ctx, L.xyz, L.xyz, ws->B, L.nc, L.nc, RK, L.neigh, stream))) return rc;
        if ((rc = psg_knn_points(ws->ctx, sub, L.xyz, ws->B, L.nc_sub, L.nc, 1, L.up, stream))) return rc;
        if (ws->B > 1) {    // per-cloud indices -> row numbers
            hipLaunchKernelGGL(offset_idx_kernel, dim3(blocks_for((size_t)L.n * RK)), dim3(256), 0, st, L.neigh, (size_t)L.nc * RK, L.nc,
                               (size_t)L.n * RK);
            PSG_LAUNCH_CHECK();
            hipLaunchKernelGGL(offset_idx_kernel, dim3(blocks_for((size_t)L.n)), dim3(256), 0, st, L.up, (size_t)L.nc, L.nc_sub, (size_t)L.n);
            PSG_LAUNCH_CHECK();
        }
        const size_t E = (size_t)L.n * RK;
        hipLaunchKernelGGL(relpos_kernel, dim3(blocks_for(E)), dim3(256), 0, st, L.xyz, L.neigh, E, L.relpos);
        PSG_LAUNCH_CHECK();
        if (ws->use_inv) {
            if ((rc = build_inverse(ws, L.neigh, E, L.n, L.inv_off, L.inv_ent, st))) return rc;
            if ((rc = build_inverse(ws, L.up, (size_t)L.n, L.n_sub, L.invu_off, L.invu_ent, st))) return rc;
            // (keys of the sampled edges: behind the builder's own two arrays in scratch_a)
            const size_t ms = (size_t)L.n_sub * RK;
            int32_t *keys_p = (int32_t *)ws->scratch_a + 2 * E;
            if ((2 * E + ms) * sizeof(int32_t) > ws->scratch_bytes) { set_error("build_pyramid: scratch too small for the sampled-edge keys"); return PSG_ERR_STATE; }
            hipLaunchKernelGGL(sampled_keys_kernel, dim3(blocks_for(ms)), dim3(256), 0, st, L.neigh, L.nc_sub, L.nc, ms, keys_p);
            PSG_LAUNCH_CHECK();
            if ((rc = build_inverse(ws, keys_p, ms, L.n, L.invp_off, L.invp_ent, st))) return rc;
        }
    }
    ws->cloud_set = true;
    ws->have_fwd = false;
    ws->xyz_branch_model = 0;
    return PSG_OK;
}

extern "C" int psg_rla_set_cloud(psg_rla_ws *ws, const float *xyz, psg_stream stream)
{
    PSG_REQUIRE(ws && xyz, "psg_rla_set_cloud: null argument");
    PSG_CHECK_HIP(hipMemcpyAsync(ws->xyz_all, xyz, (size_t)ws->N * 3 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return build_pyramid(ws, stream);
}

extern "C" const int32_t *psg_rla_index_ptr(const psg_rla_ws *ws, int what, int level)
{
    if (!ws || level < 0 || level >= RL) return nullptr;
    return what == 0 ? ws->lv[level].neigh : (what == 1 ? ws->lv[level].up : nullptr);
}

extern "C" int psg_rla_forward(psg_rla_model *m, psg_rla_ws *ws, const float *features, float *logits_out, psg_stream stream)
{
    ProfBind bind(ws);
    PSG_REQUIRE(m && ws && features && logits_out, "psg_rla_forward: null argument");
    if (!ws->cloud_set) { set_error("psg_rla_forward: psg_rla_set_cloud has not been called"); return PSG_ERR_STATE; }
    hipStream_t st = (hipStream_t)stream;
    const int N = ws->N;
    int rc;
    if ((rc = conv_fwd(m->fc0, features, 6, ws->f0, 8, N, true, ws->m_f0, st))) return rc;
    const float *fin = ws->f0;
    const bool xyz_ready = ws->xyz_branch_model == m->gen;
    for (int i = 0; i < RL; ++i) {
        LevelBuf &L = ws->lv[i];
        const EncLayers &E = m->enc[i];
        const int n = L.n, d = L.d, h = L.h;
        const size_t ne = (size_t)n * RK;
        if ((rc = conv_fwd(E.mlp1, fin, L.d_in, L.fpc, h, n, true, L.m_fpc, st))) return rc;
        if (!xyz_ready) {   // the xyz branch depends on the cloud and the weights only: once per (cloud, model)
            if ((rc = conv_fwd(E.lfa_mlp1, L.relpos, 10, L.fxyz1, h, (int)ne, true, nullptr, st))) return rc;
            if ((rc = conv_fwd(E.lfa_mlp2, L.fxyz1, h, L.fxyz2, h, (int)ne, true, nullptr, st))) return rc;
            if (ws->split && d > 16) {   // S2 = fxyz . W2^T (the position-encoding half of the score layer), kept in the cat buffers
                GemmArgs g1 = rl_args(L.fxyz1, h, E.att1_fc.w + h, d, L.cat1, d, (int)ne, h, d);
                if ((rc = rl_gemm<EPI_LINEAR>(g1, st))) return rc;
                GemmArgs g2 = rl_args(L.fxyz2, h, E.att2_fc.w + h, d, L.cat2, d, (int)ne, h, d);
                if ((rc = rl_gemm<EPI_LINEAR>(g2, st))) return rc;
            }
        }
        const bool fused16 = d == 16 && ws->fuse16;     // level 0: one kernel per attentive pooling (lfa16_*_kernel)
        const bool split = d > 16 && ws->split;         // levels 1-4: scores = T[neigh] + S2 (att_pool_split_*_kernel)
        if (split) {
            float *T1 = L.a1;                            // [n][d], kept for the backward (the attention weights are recomputed there from T + S2)
            GemmArgs g = rl_args(L.fpc, h, E.att1_fc.w, d, T1, d, n, h, d);
            g.bias = E.att1_fc.b;
            if ((rc = rl_gemm<EPI_LINEAR>(g, st))) return rc;
            {
                EvScope prof(tl_prof, 4, 0.0, st, 4.0 * (2.0 * n * d + (double)n * h + (double)ne * (1.0 + 2.0 * d + h)));
                hipLaunchKernelGGL(att_pool_split_fwd_kernel, dim3(blocks_for((size_t)n * d / 2)), dim3(256), 0, st, L.fpc, L.neigh, L.fxyz1, T1,
                                   L.cat1, h, (size_t)n * d / 2, (float *)nullptr, L.agg1);
            }
            PSG_LAUNCH_CHECK();
        } else if (fused16) {
            {
                EvScope prof(tl_prof, 6, 0.0, st, 4.0 * ((double)n * (h + d) + (double)ne * (1.0 + h)));
                hipLaunchKernelGGL(lfa16_fwd_kernel, dim3(blocks_for((size_t)n * 8)), dim3(256), 0, st, L.fpc, L.neigh, L.fxyz1, E.att1_fc.w,
                                   E.att1_fc.b, (size_t)n, L.agg1);
            }
            PSG_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(gather_concat_kernel, dim3(blocks_for(ne * d / 4)), dim3(256), 0, st, L.fpc, L.neigh, L.fxyz1, h, ne * d / 4, L.cat1);
            PSG_LAUNCH_CHECK();
            if ((rc = conv_fwd(E.att1_fc, L.cat1, d, L.a1, d, (int)ne, false, nullptr, st))) return rc;
            hipLaunchKernelGGL(att_pool_fwd_kernel, dim3(blocks_for((size_t)n * d / 4)), dim3(256), 0, st, L.cat1, L.a1, d, (size_t)n * d / 4, L.agg1);
            PSG_LAUNCH_CHECK();
        }
        if ((rc = conv_fwd(E.att1_mlp, L.agg1, d, L.fagg1, h, n, true, L.m_fagg1, st))) return rc;
        if (split) {
            float *T2 = L.a2;                            // [n][d], kept for the backward like T1
            GemmArgs g = rl_args(L.fagg1, h, E.att2_fc.w, d, T2, d, n, h, d);
            g.bias = E.att2_fc.b;
            if ((rc = rl_gemm<EPI_LINEAR>(g, st))) return rc;
            {
                EvScope prof(tl_prof, 4, 0.0, st, 4.0 * (2.0 * n * d + (double)n * h + (double)ne * (1.0 + 2.0 * d + h)));
                hipLaunchKernelGGL(att_pool_split_fwd_kernel, dim3(blocks_for((size_t)n * d / 2)), dim3(256), 0, st, L.fagg1, L.neigh, L.fxyz2, T2,
                                   L.cat2, h, (size_t)n * d / 2, (float *)nullptr, L.agg2);
            }
            PSG_LAUNCH_CHECK();
        } else if (fused16) {
            {
                EvScope prof(tl_prof, 6, 0.0, st, 4.0 * ((double)n * (h + d) + (double)ne * (1.0 + h)));
                hipLaunchKernelGGL(lfa16_fwd_kernel, dim3(blocks_for((size_t)n * 8)), dim3(256), 0, st, L.fagg1, L.neigh, L.fxyz2, E.att2_fc.w,
                                   E.att2_fc.b, (size_t)n, L.agg2);
            }
            PSG_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(gather_concat_kernel, dim3(blocks_for(ne * d / 4)), dim3(256), 0, st, L.fagg1, L.neigh, L.fxyz2, h, ne * d / 4, L.cat2);
            PSG_LAUNCH_CHECK();
            if ((rc = conv_fwd(E.att2_fc, L.cat2, d, L.a2, d, (int)ne, false, nullptr, st))) return rc;
            hipLaunchKernelGGL(att_pool_fwd_kernel, dim3(blocks_for((size_t)n * d / 4)), dim3(256), 0, st, L.cat2, L.a2, d, (size_t)n * d / 4, L.agg2);
            PSG_LAUNCH_CHECK();
        }
        if ((rc = conv_fwd(E.att2_mlp, L.agg2, d, L.fagg2, d, n, true, L.m_fagg2, st))) return rc;
        // enc = leaky_relu(mlp2(fagg2) + shortcut(fin)) (dilated_res_block, RandLANet.py:330): the shortcut's rows are the second
        // operand of mlp2's pre-activation sum (round 5: (acc + bias) + shortcut, the order of the separate add pass it replaces)
        if ((rc = conv_fwd(E.shortcut, fin, L.d_in, L.sc, 2 * d, n, false, nullptr, st))) return rc;
        {
            GemmArgs g = rl_args(L.fagg2, d, E.mlp2.w, E.mlp2.cin, L.enc, 2 * d, n, E.mlp2.cin, E.mlp2.cout);
            g.bias = E.mlp2.b; g.mask_out = L.m_enc; g.pre_add = L.sc; g.ld_pre = 2 * d;
            if ((rc = rl_gemm<EPI_LRELU>(g, st))) return rc;
        }
        hipLaunchKernelGGL(pool_max_fwd_kernel, dim3(blocks_for((size_t)L.n_sub * 2 * d / 4)), dim3(256), 0, st, L.enc, L.neigh, 2 * d,
                           (size_t)L.n_sub * 2 * d / 4, L.nc_sub, L.nc, L.samp, L.arg);
        PSG_LAUNCH_CHECK();
        fin = L.samp;
    }
    const int n5 = ws->lv[RL - 1].n_sub;
    if ((rc = conv_fwd(m->decoder0, ws->lv[RL - 1].samp, 1024, ws->dec0, 1024, n5, true, ws->m_dec0, st))) return rc;
    const float *feat = ws->dec0;
    int cfeat = 1024;
    for (int j = 0; j < RL; ++j) {
        const LevelBuf &L = ws->lv[RL - 1 - j];
        const float *skip = j == RL - 1 ? ws->lv[0].enc : ws->lv[RL - 2 - j].samp;
        const int cs = m->dec[j].cout;
        hipLaunchKernelGGL(interp_concat_kernel, dim3(blocks_for((size_t)L.n * (cs + cfeat) / 4)), dim3(256), 0, st, skip, cs, feat, cfeat,
                           L.up, (size_t)L.n * (cs + cfeat) / 4, ws->dec_cat[j]);
        PSG_LAUNCH_CHECK();
        if ((rc = conv_fwd(m->dec[j], ws->dec_cat[j], cs + cfeat, ws->dec_out[j], cs, L.n, true, ws->m_dec[j], st))) return rc;
        feat = ws->dec_out[j];
        cfeat = cs;
    }
    if ((rc = conv_fwd(m->fc1, feat, cfeat, ws->fc1o, 64, N, true, ws->m_fc1, st))) return rc;
    if ((rc = conv_fwd(m->fc2, ws->fc1o, 64, ws->fc2o, 32, N, true, ws->m_fc2, st))) return rc;
    if ((rc = conv_fwd(m->fc, ws->fc2o, 32, ws->logits, RNCLS, N, false, nullptr, st))) return rc;
    if (logits_out != ws->logits)
        PSG_CHECK_HIP(hipMemcpyAsync(logits_out, ws->logits, (size_t)N * RNCLS * 4, hipMemcpyDeviceToDevice, st));
    ws->have_fwd = true;
    ws->xyz_branch_model = m->gen;
    return PSG_OK;
}

extern "C" int psg_rla_backward(psg_rla_model *m, psg_rla_ws *ws, const float *dlogits, float *dfeatures_out, psg_stream stream)
{
    ProfBind bind(ws);
    PSG_REQUIRE(m && ws && dlogits && dfeatures_out, "psg_rla_backward: null argument");
    if (!ws->have_fwd) { set_error("psg_rla_backward: no forward is resident in the workspace"); return PSG_ERR_STATE; }
    hipStream_t st = (hipStream_t)stream;
    const int N = ws->N;
    int rc;
    // head
    // (leaky-ReLU derivatives: applied by the LAST contributor of each gradient buffer as it stores - a GEMM epilogue or an
    // inverse-list gather - instead of 29 separate passes per iteration; `fuse` is off on the atomics path, whose scatter
    // kernels cannot know when a buffer is complete)
    const bool fuse = ws->use_inv;
    if ((rc = conv_bwd(m->fc, dlogits, RNCLS, ws->d_fc2o, 32, N, 0, st, ws->m_fc2))) return rc;
    if ((rc = conv_bwd(m->fc2, ws->d_fc2o, 32, ws->d_fc1o, 64, N, 0, st, ws->m_fc1))) return rc;
    // every gradient accumulator starts at zero and collects its consumers (one memset for the whole block; the last
    // decoder layer's gradient, written next by a plain GEMM store, lies in it too)
    PSG_CHECK_HIP(hipMemsetAsync(ws->acc, 0, ws->acc_bytes, st));
    if ((rc = conv_bwd(m->fc1, ws->d_fc1o, 64, ws->d_dec_out[RL - 1], m->fc1.cin, N, 0, st, ws->m_dec[RL - 1]))) return rc;
    const int n5 = ws->lv[RL - 1].n_sub;
    // decoder, last layer first
    for (int j = RL - 1; j >= 0; --j) {
        const LevelBuf &L = ws->lv[RL - 1 - j];
        const int cs = m->dec[j].cout, cfeat = m->dec[j].cin - cs;
        float *dout = ws->d_dec_out[j];
        if (!fuse && j < RL - 1 && (rc = lrelu_bwd(dout, cs, ws->m_dec[j], L.n, cs, st))) return rc;   // (j = RL - 1: done by fc1's GEMM)
        if ((rc = conv_bwd(m->dec[j], dout, cs, ws->scratch_a, cs + cfeat, L.n, 0, st))) return rc;
        float *dskip = j == RL - 1 ? ws->lv[0].d_enc : ws->lv[RL - 2 - j].d_samp;
        float *dcoarse = j == 0 ? ws->d_dec0 : ws->d_dec_out[j - 1];
        if (ws->use_inv) {
            hipLaunchKernelGGL(interp_skip_bwd_kernel, dim3(blocks_for((size_t)L.n * cs / 4)), dim3(256), 0, st, ws->scratch_a, cs, cfeat,
                               (size_t)L.n * cs / 4, dskip);
            PSG_LAUNCH_CHECK();
            // (the coarser buffer is complete after this gather: it applies the derivative of the activation that produced it)
            hipLaunchKernelGGL(interp_inv_kernel, dim3(blocks_for((size_t)L.n_sub * cfeat)), dim3(256), 0, st, ws->scratch_a, cs, cfeat,
                               L.invu_off, L.invu_ent, (size_t)L.n_sub * cfeat, dcoarse, j == 0 ? ws->m_dec0 : ws->m_dec[j - 1]);
        } else {
            hipLaunchKernelGGL(interp_concat_bwd_kernel, dim3(blocks_for((size_t)L.n * (cs + cfeat))), dim3(256), 0, st, ws->scratch_a, cs,
                               cfeat, L.up, (size_t)L.n * (cs + cfeat), dskip, dcoarse);
        }
        PSG_LAUNCH_CHECK();
    }
    if (!fuse && (rc = lrelu_bwd(ws->d_dec0, 1024, ws->m_dec0, n5, 1024, st))) return rc;
    if ((rc = conv_bwd(m->decoder0, ws->d_dec0, 1024, ws->lv[RL - 1].d_samp, 1024, n5, 1, st))) return rc;
    // encoder, deepest level first
    for (int i = RL - 1; i >= 0; --i) {
        LevelBuf &L = ws->lv[i];
        const EncLayers &E = m->enc[i];
        const int n = L.n, d = L.d, h = L.h;
        const size_t ne = (size_t)n * RK;
        float *din = i == 0 ? ws->d_f0 : ws->lv[i - 1].d_samp;    // gradient of this level's input features
        if (ws->use_inv)
            hipLaunchKernelGGL(pool_max_inv4_kernel, dim3(blocks_for((size_t)n * 2 * d / 4)), dim3(256), 0, st, L.d_samp, L.arg, 2 * d, L.invp_off,
                               L.invp_ent, (size_t)n * 2 * d / 4, L.d_enc, L.m_enc);
        else
            hipLaunchKernelGGL(pool_max_bwd_kernel, dim3(blocks_for((size_t)L.n_sub * 2 * d)), dim3(256), 0, st, L.d_samp, L.neigh, L.arg,
                               2 * d, (size_t)L.n_sub * 2 * d, L.nc_sub, L.nc, L.d_enc);
        PSG_LAUNCH_CHECK();
        if (!fuse && (rc = lrelu_bwd(L.d_enc, 2 * d, L.m_enc, n, 2 * d, st))) return rc;
        if ((rc = conv_bwd(E.shortcut, L.d_enc, 2 * d, din, L.d_in, n, i == 0 ? 0 : 1, st))) return rc;
        float *g_fagg2 = L.agg2;   // forward buffers that are dead by now serve as gradient buffers of the same shape
        if ((rc = conv_bwd(E.mlp2, L.d_enc, 2 * d, g_fagg2, d, n, 0, st, L.m_fagg2))) return rc;
        float *g_agg2 = L.fagg2;
        if ((rc = conv_bwd(E.att2_mlp, g_fagg2, d, g_agg2, d, n, 0, st))) return rc;
        const bool fused16 = d == 16 && ws->fuse16;
        const bool split = d > 16 && ws->split;
        // split levels: ds [E][d] and the direct term [E][h] per edge, summed over every point's in-edges (dT [n][d], and
        // straight into the feature gradient), then the score layer's feature half transposed on POINTS: df += dT . W1
        auto split_bwd = [&](const float *fin_, const float *fxyz_, const float *T_, const float *S2_, const float *dagg_, const RLayer &fc,
                             float *df_, const uint32_t *mask_) -> int {
            float *ds = ws->scratch_b, *ddir = ws->scratch_a, *dT = L.m2;
            {
                const double ne_ = 16.0 * n;
                EvScope prof(tl_prof, 5, 0.0, st, 4.0 * ((double)n * (h + d) + ne_ * (1.0 + 2.0 * d + 2.0 * h)));
                hipLaunchKernelGGL(att_pool_split_bwd_kernel, dim3(blocks_for((size_t)n * d)), dim3(256), 0, st, fin_, L.neigh, fxyz_, T_, S2_, dagg_,
                                   h, (size_t)n * d, ds, ddir);
            }
            PSG_LAUNCH_CHECK();
            hipLaunchKernelGGL(gather_inv_kernel<true>, dim3(blocks_for((size_t)n * d)), dim3(256), 0, st, ds, d, d, L.inv_off, L.inv_ent,
                               (size_t)n * d, dT);
            PSG_LAUNCH_CHECK();
            hipLaunchKernelGGL(gather_inv_kernel<false>, dim3(blocks_for((size_t)n * h)), dim3(256), 0, st, ddir, h, h, L.inv_off, L.inv_ent,
                               (size_t)n * h, df_);
            PSG_LAUNCH_CHECK();
            GemmArgs g = rl_args(dT, d, fc.wt, d, df_, h, n, d, h);     // rows i < h of W^T: df[.][i] += sum_c dT[.][c] W[c][i]
            g.accumulate = 1;
            g.post_mask = mask_;                                        // last contributor of df_: the activation's derivative
            g.post_slope = kSlope;
            return rl_gemm<EPI_LINEAR>(g, st);
        };
        if (split) {
            if ((rc = split_bwd(L.fagg1, L.fxyz2, L.a2, L.cat2, g_agg2, E.att2_fc, L.d_fagg1, L.m_fagg1))) return rc;
        } else if (fused16) {
            {
                EvScope prof(tl_prof, 7, 0.0, st, 4.0 * ((double)n * (2.0 * h + d) + 16.0 * n * (1.0 + h + 8.0)));
                hipLaunchKernelGGL(lfa16_bwd_kernel, dim3(blocks_for((size_t)n * 8)), dim3(256), 0, st, L.fagg1, L.neigh, L.fxyz2, E.att2_fc.w,
                                   E.att2_fc.b, g_agg2, (size_t)n, L.d_fagg1, ws->use_inv ? ws->scratch_a : (float *)nullptr);
            }
            PSG_LAUNCH_CHECK();
            if (ws->use_inv) {
                hipLaunchKernelGGL(gather_inv_kernel<false>, dim3(blocks_for((size_t)n * h)), dim3(256), 0, st, ws->scratch_a, 8, h, L.inv_off, L.inv_ent,
                                   (size_t)n * h, L.d_fagg1, L.m_fagg1);
                PSG_LAUNCH_CHECK();
            }
        } else {
            hipLaunchKernelGGL(att_pool_bwd_kernel, dim3(blocks_for((size_t)n * d / 2)), dim3(256), 0, st, L.cat2, L.a2, g_agg2, d, (size_t)n * d / 2,
                               ws->scratch_a, ws->scratch_b);
            PSG_LAUNCH_CHECK();
            if ((rc = conv_bwd(E.att2_fc, ws->scratch_b, d, ws->scratch_a, d, (int)ne, 1, st))) return rc;
            if (ws->use_inv)
                hipLaunchKernelGGL(gather_inv_kernel<false>, dim3(blocks_for((size_t)n * h)), dim3(256), 0, st, ws->scratch_a, 2 * h, h, L.inv_off,
                                   L.inv_ent, (size_t)n * h, L.d_fagg1, L.m_fagg1);
            else
                hipLaunchKernelGGL(gather_bwd_kernel, dim3(blocks_for(ne * h)), dim3(256), 0, st, ws->scratch_a, L.neigh, h, ne * h, L.d_fagg1);
            PSG_LAUNCH_CHECK();
        }
        if (!fuse && (rc = lrelu_bwd(L.d_fagg1, h, L.m_fagg1, n, h, st))) return rc;
        float *g_agg1 = L.agg1;
        if ((rc = conv_bwd(E.att1_mlp, L.d_fagg1, h, g_agg1, d, n, 0, st))) return rc;
        if (split) {
            if ((rc = split_bwd(L.fpc, L.fxyz1, L.a1, L.cat1, g_agg1, E.att1_fc, L.d_fpc, L.m_fpc))) return rc;
        } else if (fused16) {
            {
                EvScope prof(tl_prof, 7, 0.0, st, 4.0 * ((double)n * (2.0 * h + d) + 16.0 * n * (1.0 + h + 8.0)));
                hipLaunchKernelGGL(lfa16_bwd_kernel, dim3(blocks_for((size_t)n * 8)), dim3(256), 0, st, L.fpc, L.neigh, L.fxyz1, E.att1_fc.w,
                                   E.att1_fc.b, g_agg1, (size_t)n, L.d_fpc, ws->use_inv ? ws->scratch_a : (float *)nullptr);
            }
            PSG_LAUNCH_CHECK();
            if (ws->use_inv) {
                hipLaunchKernelGGL(gather_inv_kernel<false>, dim3(blocks_for((size_t)n * h)), dim3(256), 0, st, ws->scratch_a, 8, h, L.inv_off, L.inv_ent,
                                   (size_t)n * h, L.d_fpc, L.m_fpc);
                PSG_LAUNCH_CHECK();
            }
        } else {
            hipLaunchKernelGGL(att_pool_bwd_kernel, dim3(blocks_for((size_t)n * d / 2)), dim3(256), 0, st, L.cat1, L.a1, g_agg1, d, (size_t)n * d / 2,
                               ws->scratch_a, ws->scratch_b);
            PSG_LAUNCH_CHECK();
            if ((rc = conv_bwd(E.att1_fc, ws->scratch_b, d, ws->scratch_a, d, (int)ne, 1, st))) return rc;
            if (ws->use_inv)
                hipLaunchKernelGGL(gather_inv_kernel<false>, dim3(blocks_for((size_t)n * h)), dim3(256), 0, st, ws->scratch_a, 2 * h, h, L.inv_off,
                                   L.inv_ent, (size_t)n * h, L.d_fpc, L.m_fpc);
            else
                hipLaunchKernelGGL(gather_bwd_kernel, dim3(blocks_for(ne * h)), dim3(256), 0, st, ws->scratch_a, L.neigh, h, ne * h, L.d_fpc);
            PSG_LAUNCH_CHECK();
        }
        if (!fuse && (rc = lrelu_bwd(L.d_fpc, h, L.m_fpc, n, h, st))) return rc;
        // (level 0: this GEMM completes d_f0, the gradient of fc0's leaky-ReLU output)
        if ((rc = conv_bwd(E.mlp1, L.d_fpc, h, din, L.d_in, n, 1, st, i == 0 ? ws->m_f0 : (const uint32_t *)nullptr))) return rc;
    }
    if ((rc = conv_bwd(m->fc0, ws->d_f0, 8, dfeatures_out, 6, N, 0, st))) return rc;
    ws->have_fwd = false;   // agg / fagg buffers were reused as gradient scratch
    return PSG_OK;
}

extern "C" int psg_rla_colper_grad(const float *logits, const int32_t *labels, int n, float *dlogits, float *loss_out, psg_stream stream)
{
    PSG_REQUIRE(logits && labels && dlogits && n > 0, "psg_rla_colper_grad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (loss_out) PSG_CHECK_HIP(hipMemsetAsync(loss_out, 0, 4, st));
    hipLaunchKernelGGL(colper_grad_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, logits, labels, (const uint8_t *)nullptr, 1.0f, n,
                       dlogits, loss_out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_rla_colper_grad_masked(const float *logits, const int32_t *ys, const uint8_t *mask, float sign, int n,
                                          float *dlogits, float *loss_out, psg_stream stream)
{
    PSG_REQUIRE(logits && ys && dlogits && n > 0, "psg_rla_colper_grad_masked: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (loss_out) PSG_CHECK_HIP(hipMemsetAsync(loss_out, 0, 4, st));
    hipLaunchKernelGGL(colper_grad_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, logits, ys, mask, sign, n, dlogits, loss_out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// One BIM update of the colour half of feat [n][6] from its gradient dfeat [n][6] (bim.py:84-98): l_inf or l_2 (norms:
// 2 floats of scratch), then the clip to [0, 1].  The step of psg_rla_bim_attack as an entry of its own, for the attacks
// whose loop lives on the host because the reference reads an accuracy back every iteration (TBIM's `sr > 0.9` exit).
extern "C" int psg_rla_bim_step(float *feat, const float *dfeat, const float *ori, int n, float eps, float alpha, int l2_metric,
                                float *norms, float *delta, psg_stream stream)
{
    PSG_REQUIRE(feat && dfeat && ori && n > 0 && (!l2_metric || (norms && delta)), "psg_rla_bim_step: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)n;
    if (!l2_metric) {
        hipLaunchKernelGGL(bim_linf_kernel, dim3(blocks_for(N * 3)), dim3(256), 0, st, feat, dfeat, ori, N, alpha, eps);
        PSG_LAUNCH_CHECK();
        return PSG_OK;
    }
    hipLaunchKernelGGL(sq_norm_kernel, dim3(1), dim3(1024), 0, st, dfeat, 6, 3, N, norms);
    PSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bim_l2_delta_kernel, dim3(blocks_for(N * 3)), dim3(256), 0, st, feat, dfeat, ori, N, N, alpha, norms, delta);
    PSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sq_norm_kernel, dim3(1), dim3(1024), 0, st, delta, 3, 0, N, norms + 1);
    PSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bim_l2_apply_kernel, dim3(blocks_for(N * 3)), dim3(256), 0, st, feat, ori, delta, N, N, eps, norms + 1);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// ---- NUattack / tar_NUattack (ares/ares/attack/NUattack.py:12-74, tar_NUattack.py:12-84): Adam on d_ws in tanh space.
//   ws = atanh(2 b x - b) + d_ws (b = 1 - 1e-6), adv = (tanh(ws) + 1) / 2, masked: adv = mask adv + (1 - mask) x;
//   loss = |adv - x|_2 + c * score;  TF1 Adam: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), d_ws -= lr_t m / (sqrt(v) + 1e-8).
namespace {
constexpr float kNuBound = 1.0f - 1e-6f;

__global__ void nu_color_kernel(const float *__restrict__ xs, const float *__restrict__ dws, const uint8_t *__restrict__ mask,
                                size_t n, float *__restrict__ feat, float *__restrict__ dist2)
{
    float s = 0.0f;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * 3; t += (size_t)gridDim.x * blockDim.x) {
        const size_t i = t / 3;
        const int c = (int)(t % 3);
        const float x = xs[t];
        const float w = atanhf(2.0f * kNuBound * x - kNuBound) + dws[t];
        float adv = 0.5f * (tanhf(w) + 1.0f);
        if (mask && !mask[i]) adv = x;
        feat[i * 6 + 3 + c] = adv;
        const float d = adv - x;
        s += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(dist2, s);
}

__global__ void nu_adam_kernel(const float *__restrict__ xs, float *__restrict__ dws, float *__restrict__ m, float *__restrict__ v,
                               const uint8_t *__restrict__ mask, const float *__restrict__ feat, const float *__restrict__ dfeat,
                               const float *__restrict__ dist2, size_t n, float c, float lr_t, float b1, float b2, float eps)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 3) return;
    const size_t i = t / 3;
    const int ch = (int)(t % 3);
    const float x = xs[t];
    const float w = atanhf(2.0f * kNuBound * x - kNuBound) + dws[t];
    const float th = tanhf(w);
    const float adv = feat[i * 6 + 3 + ch];
    const float dist = sqrtf(dist2[0]);
    // d loss / d adv = (adv - x) / |adv - x|_2 + c * d score / d adv;  d adv / d d_ws = mask * (1 - tanh^2) / 2
    float g = (dist > 0.0f ? (adv - x) / dist : 0.0f) + c * dfeat[i * 6 + 3 + ch];
    g *= (mask && !mask[i]) ? 0.0f : 0.5f * (1.0f - th * th);
    const float mm = b1 * m[t] + (1.0f - b1) * g;
    const float vv = b2 * v[t] + (1.0f - b2) * g * g;
    m[t] = mm;
    v[t] = vv;
    dws[t] -= lr_t * mm / (sqrtf(vv) + eps);
}
}  // namespace

// adv colours of d_ws into feat [n][6] (columns 3..5) and dist2[0] = |adv - x|_2^2 (cleared here)
extern "C" int psg_rla_nu_color(const float *xs, const float *dws, const uint8_t *mask, int n, float *feat, float *dist2,
                                psg_stream stream)
{
    PSG_REQUIRE(xs && dws && feat && dist2 && n > 0, "psg_rla_nu_color: bad argument");
    hipStream_t st = (hipStream_t)stream;
    PSG_CHECK_HIP(hipMemsetAsync(dist2, 0, 4, st));
    hipLaunchKernelGGL(nu_color_kernel, dim3(256), dim3(256), 0, st, xs, dws, mask, (size_t)n, feat, dist2);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// one Adam step (t = 1, 2, ..) on d_ws from dfeat = d score / d features [n][6] of the forward that psg_rla_nu_color fed
extern "C" int psg_rla_nu_adam_step(const float *xs, float *dws, float *m, float *v, const uint8_t *mask, const float *feat,
                                    const float *dfeat, const float *dist2, int n, float c, float lr, int t, psg_stream stream)
{
    PSG_REQUIRE(xs && dws && m && v && feat && dfeat && dist2 && n > 0 && t >= 1, "psg_rla_nu_adam_step: bad argument");
    const double b1 = 0.9, b2 = 0.999;
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t)));
    hipLaunchKernelGGL(nu_adam_kernel, dim3(blocks_for((size_t)n * 3)), dim3(256), 0, (hipStream_t)stream, xs, dws, m, v, mask, feat,
                       dfeat, dist2, (size_t)n, c, lr_t, 0.9f, 0.999f, 1e-8f);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// BIM (ares/ares/attack/bim.py:190-236): `iters` gradient steps on the colour half of the features of one cloud.
extern "C" int psg_rla_bim_attack(psg_rla_model *m, psg_rla_ws *ws, const float *features, const int32_t *labels, float eps,
                                  float alpha, int iters, int l2_metric, float *adv_features_out, psg_stream stream)
{
    PSG_REQUIRE(m && ws && features && labels && adv_features_out && iters > 0, "psg_rla_bim_attack: bad argument");
    ProfBind bind(ws);
    hipStream_t st = (hipStream_t)stream;
    const size_t N = ws->N;
    int rc;
    PSG_CHECK_HIP(hipMemcpyAsync(ws->feat, features, N * 6 * 4, hipMemcpyDeviceToDevice, st));
    PSG_CHECK_HIP(hipMemcpyAsync(ws->labels, labels, N * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks_for(N * 3)), dim3(256), 0, st, features, 6, 3, 3, N, ws->ori, 3, 0);
    PSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks_for(N * 3)), dim3(256), 0, st, features, 6, 0, 3, N, ws->xyz_all, 3, 0);
    PSG_LAUNCH_CHECK();
    if ((rc = build_pyramid(ws, stream))) return rc;        // geometry: once per cloud, the attack moves colours only
    auto iteration = [&]() -> int {
        int r;
        if ((r = psg_rla_forward(m, ws, ws->feat, ws->logits, stream))) return r;
        if ((r = psg_rla_colper_grad(ws->logits, ws->labels, (int)N, ws->dlogits, nullptr, stream))) return r;
        if ((r = psg_rla_backward(m, ws, ws->dlogits, ws->dfeat, stream))) return r;
        if (!l2_metric) {
            hipLaunchKernelGGL(bim_linf_kernel, dim3(blocks_for(N * 3)), dim3(256), 0, st, ws->feat, ws->dfeat, ws->ori, N, alpha, eps);
            PSG_LAUNCH_CHECK();
        } else {
            // per cloud (bim.py:84-98 normalises every sample of the batch by its own norms): norms [2][B]
            const size_t Nc = (size_t)ws->Nc;
            hipLaunchKernelGGL(sq_norm_kernel, dim3(ws->B), dim3(1024), 0, st, ws->dfeat, 6, 3, Nc, ws->norms);
            PSG_LAUNCH_CHECK();
            hipLaunchKernelGGL(bim_l2_delta_kernel, dim3(blocks_for(N * 3)), dim3(256), 0, st, ws->feat, ws->dfeat, ws->ori, N, Nc, alpha,
                               ws->norms, ws->delta);
            PSG_LAUNCH_CHECK();
            hipLaunchKernelGGL(sq_norm_kernel, dim3(ws->B), dim3(1024), 0, st, ws->delta, 3, 0, Nc, ws->norms + ws->B);
            PSG_LAUNCH_CHECK();
            hipLaunchKernelGGL(bim_l2_apply_kernel, dim3(blocks_for(N * 3)), dim3(256), 0, st, ws->feat, ws->ori, ws->delta, N, Nc, eps,
                               ws->norms + ws->B);
            PSG_LAUNCH_CHECK();
        }
        return PSG_OK;
    };
    // The first iteration runs eagerly (it computes the xyz branch and sets kernel attributes outside any capture); the
    // others enqueue the same launches with the same arguments, so they are captured once into a hipGraph kept in the
    // workspace and replayed: one graph launch instead of ~150 kernel launches per iteration, which is what lets several
    // long attacks on different streams actually overlap (the host would otherwise spend its time filling one stream's
    // queue).  Capture is impossible on the legacy default stream; the loop then stays eager.
    if ((rc = iteration())) return rc;
    int it = 1;
    static const bool use_graph = !((psg::env_int("PSG_RLA_NO_GRAPH", 0) != 0)) && !trace_sync_enabled();   // (the tracer synchronises after every launch)
    if (use_graph && !ws->prof.on && iters - it >= 2) {
        const bool same_key = ws->bim_model_gen == m->gen && ws->bim_eps == eps && ws->bim_alpha == alpha && ws->bim_metric == l2_metric;
        if (!same_key) {
            if (ws->bim_exec) {
                PSG_CHECK_HIP(hipStreamSynchronize(st));
                (void)hipGraphExecDestroy(ws->bim_exec);
                ws->bim_exec = nullptr;
            }
            ws->bim_capture_failed = false;
        }
        if (!ws->bim_exec && !ws->bim_capture_failed) {
            // a capture that fails (refused on the legacy stream, or invalidated) has executed nothing: the iterations run
            // eagerly below - and show a genuine launch error there -, the failure is counted (psg_capture_stats) and this
            // key is not tried again
            bool ok = false;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int crc = iteration();
                hipGraph_t graph = nullptr;
                const hipError_t e = hipStreamEndCapture(st, &graph);
                ok = crc == PSG_OK && e == hipSuccess && graph && hipGraphInstantiate(&ws->bim_exec, graph, nullptr, nullptr, 0) == hipSuccess;
                if (!ok) ws->bim_exec = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
            ws->bim_model_gen = m->gen; ws->bim_eps = eps; ws->bim_alpha = alpha; ws->bim_metric = l2_metric;
            ws->bim_capture_failed = !ok;
            psg::capture_note(&ws->cap, 1, ok ? 0 : 1, 0, 0);
        }
        if (ws->bim_exec) {
            psg::capture_note(&ws->cap, 0, 0, iters - it, 0);
            for (; it < iters; ++it) PSG_CHECK_HIP(hipGraphLaunch(ws->bim_exec, st));
        } else {
            psg::capture_note(&ws->cap, 0, 0, 0, iters - it);
        }
    }
    for (; it < iters; ++it)
        if ((rc = iteration())) return rc;
    PSG_CHECK_HIP(hipMemcpyAsync(adv_features_out, ws->feat, N * 6 * 4, hipMemcpyDeviceToDevice, st));
    return PSG_OK;
}
