// Fused set-abstraction / feature-propagation kernels (forward and input-gradient backward) for
// PointNet++ SSG on gfx950.  One workgroup owns P points (P/32 ball-query groups for SA); the whole
// MLP chain of the module runs out of LDS, only module inputs/outputs, ReLU bit-masks and max-pool
// arg-indices touch HBM.
//
// Reference semantics (paths relative to /root/reference):
//   SA  forward  PointNet/models/pointnet_util.py:126-140 (gather, grouped_xyz - new_xyz, concat
//                [rel_xyz, feats]) + :200-205 (3 x conv1x1+BN+ReLU, max over the 32 samples)
//   FP  forward  pointnet_util.py:308-319 (3-NN weighted sum, concat [points1, interp], conv1x1+BN+ReLU)
//   head         PointNet/models/pointnet2_sem_seg.py:36-38 (conv1+bn1+ReLU, conv2, log_softmax)
//   backward     what autograd derives for d/d(input features); geometry is constant w.r.t. colour.
#pragma once
#include "psg_mlp.cuh"

namespace psg {

struct SaFwdArgs {
    const float *xyz;      // [B][Np][xyz_stride], first 3 floats of a row = xyz
    const float *feat;     // [B][Np][D]
    const float *new_xyz;  // [B][S][3]
    const int32_t *gidx;   // [B][S][32]
    float *out;            // [B][S][C3]
    uint8_t *arg;          // [B][S][C3] arg-max sample, 255 = no gradient (max <= 0)
    FwdLayer l1, l2;
    const float4 *w3;      // last layer, packed like FwdLayer::w, used as the B operand (flipped tile)
    const float *b3;
    int k8_3, nb3;
    int xyz_stride, D, Np, S, C3;
    int rows0, rows1;
};

struct SaBwdArgs {
    const float *dout;    // [B][S][C3]
    const uint8_t *arg;   // [B][S][C3]
    const int32_t *gidx;  // [B][S][32]
    float *dfeat;         // [B][Np][D]  (atomicAdd)
    BwdLayer l3t, l2t, l1t;
    int D, Np, S, C3;
    int c_lo, c_hi;       // grouped-row channels [c_lo, c_hi) are scattered to feature c-3
    int rows0, rows1;
};

struct FpFwdArgs {
    const float *feat1;     // skip features [B][N][C1] or null
    const float *feat2;     // coarse features [B][S][C2]
    const int32_t *nn_idx;  // [B][N][3]
    const float *nn_w;      // [B][N][3]
    float *out;             // [B][N][Cout] or null
    float *logp;            // [B][N][n_cls] or null: log_softmax of the last layer's first n_cls rows
    FwdLayer layer[MAX_LAYERS];
    int n_layers;
    int C1, C2, N, S, Cout, n_cls;
    int rows0, rows1;
};

struct FpBwdArgs {
    const float *dout;          // [B][N][Cout] gradient of the last layer's (post-ReLU) output, or null
    const uint16_t *mask_last;  // ReLU mask of that layer
    const float *logp;          // head mode: [B][N][n_cls]
    const float *dlogp;         // head mode: [B][N][n_cls]
    const int32_t *nn_idx;
    const float *nn_w;
    float *dfeat1;              // [B][N][C1] (atomicAdd) or null
    float *dfeat2;              // [B][S][C2] (atomicAdd)
    BwdLayer layer[MAX_LAYERS];
    int n_layers;
    int C1, C2, N, S, Cout, n_cls, mb_last;
    int rows0, rows1;
};

// ------------------------------------------------------------------------------------------ SA fwd
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void sa_fwd_kernel(SaFwdArgs a)
{
    constexpr int LDP = P + 1, G = P / 32, NT = NW * 64, NPART = NT / P;
    extern __shared__ float lds[];
    float *buf0 = lds, *buf1 = lds + (size_t)a.rows0 * LDP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, s0 = blockIdx.x * G;
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;

    {   // gather [rel_xyz(3), feats(D)] of the P grouped points into buf0[channel][point]
        const int j = tid % P, part = tid / P;
        const int s = s0 + (j >> 5);
        const int src = a.gidx[((size_t)b * a.S + s) * 32 + (j & 31)];
        const float *frow = a.feat + ((size_t)b * a.Np + src) * a.D;
        if (part == 0) {
            const float *xr = a.xyz + ((size_t)b * a.Np + src) * a.xyz_stride;
            const float *cr = a.new_xyz + ((size_t)b * a.S + s) * 3;
            buf0[0 * LDP + j] = xr[0] - cr[0];
            buf0[1 * LDP + j] = xr[1] - cr[1];
            buf0[2 * LDP + j] = xr[2] - cr[2];
        }
        if ((a.D & 3) == 0) {
            const float4 *f4 = (const float4 *)frow;
            for (int q = part; q < (a.D >> 2); q += NPART) {
                float4 v = f4[q];
                float *o = buf0 + (size_t)(3 + 4 * q) * LDP + j;
                o[0] = v.x; o[LDP] = v.y; o[2 * LDP] = v.z; o[3 * LDP] = v.w;
            }
        } else {
            for (int c = part; c < a.D; c += NPART) buf0[(size_t)(3 + c) * LDP + j] = frow[c];
        }
        for (int c = 3 + a.D + part; c < a.l1.k8 * 8; c += NPART) buf0[(size_t)c * LDP + j] = 0.0f;
    }
    __syncthreads();
    layer_fwd<P, NW>(a.l1, buf0, buf1, wg);
    __syncthreads();
    layer_fwd<P, NW>(a.l2, buf1, buf0, wg);
    __syncthreads();

    // last layer with the tile flipped (D[point][channel]) so the max over the 32 samples of a
    // group is an in-lane max over 16 accumulators + one exchange between lane halves.
    const int jj = lane & 31, h = lane >> 5;
    for (int task = wave; task < a.nb3 * G; task += NW) {
        const int nb = task / G, g = task - nb * G;
        f32x16 acc;
        const float bias = a.b3[nb * 32 + jj];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias;
        acc = tile_mac<LDP, true>(a.w3 + (size_t)nb * a.k8_3 * 64 + lane, a.k8_3, buf0 + h * LDP + g * 32 + jj, acc);
        float best = -1.0f;
        int bidx = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] > 0.0f ? acc[r] : 0.0f;
            if (v > best) { best = v; bidx = acc_row(r, h); }
        }
        float ob = __shfl_xor(best, 32);
        int oi = __shfl_xor(bidx, 32);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        if (h == 0) {
            size_t o = ((size_t)b * a.S + s0 + g) * a.C3 + nb * 32 + jj;
            a.out[o] = best;
            a.arg[o] = best > 0.0f ? (uint8_t)bidx : (uint8_t)255;
        }
    }
}

// ------------------------------------------------------------------------------------------ SA bwd
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void sa_bwd_kernel(SaBwdArgs a)
{
    constexpr int LDP = P + 1, G = P / 32, NT = NW * 64;
    extern __shared__ float lds[];
    float *buf0 = lds, *buf1 = lds + (size_t)a.rows0 * LDP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, s0 = blockIdx.x * G;
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;

    // max-pool backward: dZ3[c][g*32+k] = dout[g][c] if k == arg[g][c] else 0
    for (int pair = wave * 2 + (lane >> 5); pair < a.C3 * G; pair += NW * 2) {
        const int g = pair / a.C3, c = pair - g * a.C3;
        const size_t o = ((size_t)b * a.S + s0 + g) * a.C3 + c;
        const int am = a.arg[o];
        const float v = a.dout[o];
        buf0[(size_t)c * LDP + g * 32 + (lane & 31)] = ((lane & 31) == am) ? v : 0.0f;
    }
    __syncthreads();
    layer_bwd<P, NW>(a.l3t, buf0, buf1, wg);
    __syncthreads();
    layer_bwd<P, NW>(a.l2t, buf1, buf0, wg);
    __syncthreads();
    layer_bwd<P, NW>(a.l1t, buf0, buf1, wg);
    __syncthreads();
    // index_points backward: scatter-add the feature rows of the grouped-input gradient
    const int nc = a.c_hi - a.c_lo;
    for (int t = tid; t < P * nc; t += NT) {
        const int j = t / nc, c = a.c_lo + (t - j * nc);
        const float v = buf1[(size_t)c * LDP + j];
        if (v != 0.0f) {
            const int src = a.gidx[((size_t)b * a.S + s0 + (j >> 5)) * 32 + (j & 31)];
            atomicAdd(a.dfeat + ((size_t)b * a.Np + src) * a.D + (c - 3), v);
        }
    }
}

// ------------------------------------------------------------------------------------------ FP fwd
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void fp_fwd_kernel(FpFwdArgs a)
{
    constexpr int LDP = P + 1, NT = NW * 64, NPART = NT / P;
    extern __shared__ float lds[];
    float *buf0 = lds, *buf1 = lds + (size_t)a.rows0 * LDP;
    const int tid = threadIdx.x;
    const int b = blockIdx.y, n0 = blockIdx.x * P;
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    {
        const int j = tid % P, part = tid / P;
        const size_t n = (size_t)b * a.N + n0 + j;
        if (a.feat1) {
            const float4 *f4 = (const float4 *)(a.feat1 + n * a.C1);
            for (int q = part; q < (a.C1 >> 2); q += NPART) {
                float4 v = f4[q];
                float *o = buf0 + (size_t)(4 * q) * LDP + j;
                o[0] = v.x; o[LDP] = v.y; o[2 * LDP] = v.z; o[3 * LDP] = v.w;
            }
        }
        const int i0 = a.nn_idx[n * 3], i1 = a.nn_idx[n * 3 + 1], i2 = a.nn_idx[n * 3 + 2];
        const float w0 = a.nn_w[n * 3], w1 = a.nn_w[n * 3 + 1], w2 = a.nn_w[n * 3 + 2];
        const float4 *g0 = (const float4 *)(a.feat2 + ((size_t)b * a.S + i0) * a.C2);
        const float4 *g1 = (const float4 *)(a.feat2 + ((size_t)b * a.S + i1) * a.C2);
        const float4 *g2 = (const float4 *)(a.feat2 + ((size_t)b * a.S + i2) * a.C2);
        for (int q = part; q < (a.C2 >> 2); q += NPART) {
            float4 u0 = g0[q], u1 = g1[q], u2 = g2[q];
            float *o = buf0 + (size_t)(a.C1 + 4 * q) * LDP + j;
            o[0] = u0.x * w0 + u1.x * w1 + u2.x * w2;
            o[LDP] = u0.y * w0 + u1.y * w1 + u2.y * w2;
            o[2 * LDP] = u0.z * w0 + u1.z * w1 + u2.z * w2;
            o[3 * LDP] = u0.w * w0 + u1.w * w1 + u2.w * w2;
        }
    }
    __syncthreads();
    float *in = buf0, *out = buf1;
    for (int l = 0; l < a.n_layers; ++l) {
        layer_fwd<P, NW>(a.layer[l], in, out, wg);
        __syncthreads();
        float *t = in; in = out; out = t;
    }
    // `in` now holds the last layer's output [channel][point]
    if (a.out) {
        for (int t = tid; t < P * a.Cout; t += NT) {
            const int j = t / a.Cout, c = t - j * a.Cout;
            a.out[((size_t)b * a.N + n0 + j) * a.Cout + c] = in[(size_t)c * LDP + j];
        }
    }
    if (a.logp && tid < P) {
        const int j = tid;
        float z[16];
        float m = -INFINITY;
        for (int c = 0; c < a.n_cls; ++c) { z[c] = in[(size_t)c * LDP + j]; m = fmaxf(m, z[c]); }
        float s = 0.0f;
        for (int c = 0; c < a.n_cls; ++c) s += expf(z[c] - m);
        const float lse = logf(s);
        float *o = a.logp + ((size_t)b * a.N + n0 + j) * a.n_cls;
        for (int c = 0; c < a.n_cls; ++c) o[c] = (z[c] - m) - lse;
    }
}

// ------------------------------------------------------------------------------------------ FP bwd
template <int P, int NW>
__global__ __launch_bounds__(NW * 64) void fp_bwd_kernel(FpBwdArgs a)
{
    constexpr int LDP = P + 1, PB = P / 32, NT = NW * 64;
    extern __shared__ float lds[];
    float *buf0 = lds, *buf1 = lds + (size_t)a.rows0 * LDP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, n0 = blockIdx.x * P;
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;

    if (a.dout) {
        // dZ_last = dout * mask_last, loaded tile-wise so the mask bits line up with the forward epilogue
        const int j = lane & 31, h = lane >> 5;
        const int ntask = a.mb_last * PB;
        for (int task = wave; task < ntask; task += NW) {
            const int mb = task / PB, pb = task - mb * PB;
            const unsigned m = a.mask_last[(wg * ntask + task) * 64 + lane];
            const float *row = a.dout + ((size_t)b * a.N + n0 + pb * 32 + j) * a.Cout + mb * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = *(const float4 *)(row + 8 * g);
                float *o = buf0 + (size_t)(mb * 32 + 8 * g + 4 * h) * LDP + pb * 32 + j;
                o[0] = ((m >> (4 * g)) & 1u) ? v.x : 0.0f;
                o[LDP] = ((m >> (4 * g + 1)) & 1u) ? v.y : 0.0f;
                o[2 * LDP] = ((m >> (4 * g + 2)) & 1u) ? v.z : 0.0f;
                o[3 * LDP] = ((m >> (4 * g + 3)) & 1u) ? v.w : 0.0f;
            }
        }
    } else if (tid < P) {
        // log_softmax backward: dz = dlogp - exp(logp) * sum(dlogp)   (pointnet2_sem_seg.py:38)
        const int j = tid;
        const size_t n = (size_t)b * a.N + n0 + j;
        const float *lp = a.logp + n * a.n_cls, *dl = a.dlogp + n * a.n_cls;
        float s = 0.0f;
        for (int c = 0; c < a.n_cls; ++c) s += dl[c];
        for (int c = 0; c < a.n_cls; ++c) buf0[(size_t)c * LDP + j] = dl[c] - expf(lp[c]) * s;
        for (int c = a.n_cls; c < a.layer[0].k8 * 8; ++c) buf0[(size_t)c * LDP + j] = 0.0f;
    }
    __syncthreads();
    float *in = buf0, *out = buf1;
    for (int l = 0; l < a.n_layers; ++l) {
        layer_bwd<P, NW>(a.layer[l], in, out, wg);
        __syncthreads();
        float *t = in; in = out; out = t;
    }
    // `in` = gradient of the concat input [C1 skip rows | C2 interpolated rows][point]
    if (a.dfeat1) {
        for (int t = tid; t < P * a.C1; t += NT) {
            const int j = t / a.C1, c = t - j * a.C1;
            const float v = in[(size_t)c * LDP + j];
            if (v != 0.0f) atomicAdd(a.dfeat1 + ((size_t)b * a.N + n0 + j) * a.C1 + c, v);
        }
    }
    for (int t = tid; t < P * a.C2; t += NT) {
        const int j = t / a.C2, c = t - j * a.C2;
        const float v = in[(size_t)(a.C1 + c) * LDP + j];
        if (v != 0.0f) {
            const size_t n = (size_t)b * a.N + n0 + j;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                atomicAdd(a.dfeat2 + ((size_t)b * a.S + a.nn_idx[n * 3 + k]) * a.C2 + c, a.nn_w[n * 3 + k] * v);
        }
    }
}

}  // namespace psg
