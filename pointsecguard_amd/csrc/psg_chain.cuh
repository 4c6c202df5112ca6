// Wave-private MLP chains: ONE wave owns 32 points through every layer of a module whose layers are at most
// 128 channels wide (fp1 + classifier head: 128 -> 128 x4 -> 13).
//
// Why (measured, tools/mfma_probe.hip + tools/diag_*.sh): in the workgroup-cooperative kernels (psg_mlp.cuh)
// a wave owns one 32x32 tile of a layer, i.e. 64 MFMAs between two barriers; per tile it pays a weight-stream
// start-up, an epilogue and a barrier wait for the slowest of its siblings, and the matrix pipe idles ~40 % of
// the time.  Here a wave keeps ALL FOUR 32-row output tiles of a layer in its own accumulators (64 AGPRs), so
//   * one activation read (ds_read_b128) feeds 16 MFMAs instead of 4,
//   * a layer is 256 MFMAs (16k cycles) of uninterrupted issue with four independent accumulation chains,
//   * there is no barrier anywhere: the activations live in a wave-private LDS region updated in place
//     (all reads of a layer are complete before its outputs are written back), and
//   * workgroups are single waves, so the CU schedules them independently (no convoying at layer boundaries).
// Weights are packed [k8][mb = 4][64 lanes] float4 so one k8-step (8 input channels x 128 outputs) is 4 KiB
// contiguous; they stream from L2 through a two-step ring in fixed VGPRs v[64:95] with counted waits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psg_mlp.cuh"

namespace psg {

constexpr int WBLK = 32 * 8 + 8;   // floats per 8-channel block of a wave's 32 points (k8-block layout, P = 32)

struct ChainLayer {
    const float4 *w4;   // [k8][4][64] float4: lane (i,h) elem t of block mb = W[mb*32+i][8*k8+4*h+t]
    const float *bias;  // [128] or null
    int k8;             // multiple of 4
};

#define PSG_MF(acc, a, b) "v_mfma_f32_32x32x2_f32 %[" #acc "], v" #a ", v" #b ", %[" #acc "]\n\t"
#define PSG_STEP_A(b0, b1, b2, b3)                                                                              \
    PSG_MF(c0, 64, b0) PSG_MF(c1, 68, b0) PSG_MF(c2, 72, b0) PSG_MF(c3, 76, b0) PSG_MF(c0, 65, b1) PSG_MF(c1, 69, b1) \
    PSG_MF(c2, 73, b1) PSG_MF(c3, 77, b1) PSG_MF(c0, 66, b2) PSG_MF(c1, 70, b2) PSG_MF(c2, 74, b2) PSG_MF(c3, 78, b2) \
    PSG_MF(c0, 67, b3) PSG_MF(c1, 71, b3) PSG_MF(c2, 75, b3) PSG_MF(c3, 79, b3)
#define PSG_STEP_B(b0, b1, b2, b3)                                                                              \
    PSG_MF(c0, 80, b0) PSG_MF(c1, 84, b0) PSG_MF(c2, 88, b0) PSG_MF(c3, 92, b0) PSG_MF(c0, 81, b1) PSG_MF(c1, 85, b1) \
    PSG_MF(c2, 89, b1) PSG_MF(c3, 93, b1) PSG_MF(c0, 82, b2) PSG_MF(c1, 86, b2) PSG_MF(c2, 90, b2) PSG_MF(c3, 94, b2) \
    PSG_MF(c0, 83, b3) PSG_MF(c1, 87, b3) PSG_MF(c2, 91, b3) PSG_MF(c3, 95, b3)

// c[mb] += W[mb*32.., :] . act for the four 32-row blocks.  w = layer.w4 + lane, k8n a multiple of 4,
// bptr = act + (lane&31)*8 + 4*(lane>>5).  Two k8-steps per loop iteration: ring slot A = v[64:79], slot B =
// v[80:95] (4 chunks of 1 KiB per wave each), activation operands v[96:99] / v[100:103] one step ahead.
// Every wait is counted: "vmcnt(4)" = the slot about to be used has landed, the other slot may be in flight.
__device__ __forceinline__ void mac4(const float4 *w, int k8n, const float *bptr, f32x16 &c0, f32x16 &c1, f32x16 &c2,
                                     f32x16 &c3)
{
#if defined(__HIP_DEVICE_COMPILE__)   // AMDGPU inline assembly: device pass only
    unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)bptr;
    const float4 *wp = w;
    int n = (k8n >> 1) - 1;   // pipelined step pairs; the last pair runs without re-loads
    const unsigned long long step = 4096ull;
    constexpr int S1 = WBLK * 4, S2 = 2 * WBLK * 4;
    asm volatile(
        "global_load_dwordx4 v[64:67], %[wp], off\n\t"
        "global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t"
        "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
        "global_load_dwordx4 v[80:83], %[wp], off\n\t"
        "global_load_dwordx4 v[84:87], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[88:91], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[92:95], %[wp], off offset:3072\n\t"
        "ds_read_b128 v[96:99], %[lds]\n\t"
        "L_psg4_loop_%=:\n\t"
        "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
        "ds_read_b128 v[100:103], %[lds] offset:%[s1]\n\t"
        "s_waitcnt vmcnt(4) lgkmcnt(1)\n\t"
        PSG_STEP_A(96, 97, 98, 99)
        "global_load_dwordx4 v[64:67], %[wp], off\n\t"
        "global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t"
        "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
        "ds_read_b128 v[96:99], %[lds] offset:%[s2]\n\t"
        "s_waitcnt vmcnt(4) lgkmcnt(1)\n\t"
        PSG_STEP_B(100, 101, 102, 103)
        "global_load_dwordx4 v[80:83], %[wp], off\n\t"
        "global_load_dwordx4 v[84:87], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[88:91], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[92:95], %[wp], off offset:3072\n\t"
        "v_add_u32 %[lds], %[s2], %[lds]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 L_psg4_loop_%=\n\t"
        "ds_read_b128 v[100:103], %[lds] offset:%[s1]\n\t"
        "s_waitcnt vmcnt(4) lgkmcnt(1)\n\t"
        PSG_STEP_A(96, 97, 98, 99)
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        PSG_STEP_B(100, 101, 102, 103)
        "s_nop 15\n\t"
        "s_nop 3\n\t"
        : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3), [wp] "+v"(wp), [lds] "+v"(lds), [n] "+s"(n)
        : [step] "s"(step), [s1] "n"(S1), [s2] "n"(S2)
        : "memory", "scc", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
          "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91",
          "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103");
#endif
}

// accumulators <- bias rows of block mb (lane (j,h) register 4g+e = channel mb*32 + 8g + 4h + e), or zero
__device__ __forceinline__ f32x16 bias_tile(const float *bias, int mb, int h)
{
    f32x16 c;
    if (bias) {
        const float4 *bp = (const float4 *)(bias + mb * 32 + 4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = bp[2 * g];
            c[4 * g] = v.x; c[4 * g + 1] = v.y; c[4 * g + 2] = v.z; c[4 * g + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    }
    return c;
}

__device__ __forceinline__ void wave_store_tile(float *act, int mb, int j, int h, const f32x16 &v)
{
    float *o = act + (size_t)(mb * 4) * WBLK + j * 8 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *(float4 *)(o + (size_t)g * WBLK) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

// ReLU + mask bits + write-back of one tile; returns the 16 mask bits of this lane
__device__ __forceinline__ unsigned relu_tile(float *act, int mb, int j, int h, f32x16 v)
{
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool pos = v[r] > 0.0f;
        m |= (unsigned)pos << r;
        v[r] = pos ? v[r] : 0.0f;
    }
    wave_store_tile(act, mb, j, h, v);
    return m;
}

__device__ __forceinline__ void mask_tile(float *act, int mb, int j, int h, f32x16 v, unsigned m)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = ((m >> r) & 1u) ? v[r] : 0.0f;
    wave_store_tile(act, mb, j, h, v);
}

// ------------------------------------------------------------------------------------- fp1 + head, forward
struct Fp1FwdArgs {
    const float *feat2;       // [B][S][128] coarse features (fp2 output)
    const int32_t *nn_idx;    // [B][N][3]
    const float *nn_w;        // [B][N][3]
    float *logp;              // [B][N][n_cls]
    unsigned long long *mask[4];  // per 128-wide ReLU layer: [tile][64 lanes], 16 bits per 32-row block
    ChainLayer layer[4];      // fp1.mlp0..2, conv1 (BatchNorm folded)
    FwdLayer head;            // conv2, classic [mb][k8][64] packing (one 32-row tile)
    int N, S, n_cls;
};

// pointnet_util.py:308-319 (interpolate + MLP) and pointnet2_sem_seg.py:36-38 (conv1, conv2, log_softmax)
__global__ __launch_bounds__(64) void fp1_fwd_wave_kernel(Fp1FwdArgs a)
{
    extern __shared__ float act[];
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    int bx, b;
    xcd_tile(bx, b);
    const int n0 = bx * 32;
    const size_t tile = (size_t)b * gridDim.x + bx;

    // 3-NN inverse-distance interpolation of the coarse features.  Lane p < 32 fetches the neighbour indices and
    // weights of point p once (one memory latency for the whole tile); every pass then takes two points (one per
    // lane half), their indices arriving by cross-lane permute, 32 lanes x float4 = one 512-byte source row each.
    {
        const float *f2 = a.feat2 + (size_t)b * a.S * 128;
        const size_t nl = ((size_t)b * a.N + n0 + j) * 3;
        const int li0 = a.nn_idx[nl], li1 = a.nn_idx[nl + 1], li2 = a.nn_idx[nl + 2];
        const float lw0 = a.nn_w[nl], lw1 = a.nn_w[nl + 1], lw2 = a.nn_w[nl + 2];
#pragma unroll 8
        for (int p = 0; p < 32; p += 2) {
            const int pp = p + h;
            const int i0 = __shfl(li0, pp), i1 = __shfl(li1, pp), i2 = __shfl(li2, pp);
            const float w0 = __shfl(lw0, pp), w1 = __shfl(lw1, pp), w2 = __shfl(lw2, pp);
            const float4 u0 = *(const float4 *)(f2 + (size_t)i0 * 128 + 4 * j);
            const float4 u1 = *(const float4 *)(f2 + (size_t)i1 * 128 + 4 * j);
            const float4 u2 = *(const float4 *)(f2 + (size_t)i2 * 128 + 4 * j);
            float4 r;
            r.x = u0.x * w0 + u1.x * w1 + u2.x * w2;
            r.y = u0.y * w0 + u1.y * w1 + u2.y * w2;
            r.z = u0.z * w0 + u1.z * w1 + u2.z * w2;
            r.w = u0.w * w0 + u1.w * w1 + u2.w * w2;
            *(float4 *)(act + (size_t)(j >> 1) * WBLK + pp * 8 + 4 * (j & 1)) = r;
        }
    }
    const float *bptr = act + j * 8 + 4 * h;
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const ChainLayer L = a.layer[l];
        f32x16 c0 = bias_tile(L.bias, 0, h), c1 = bias_tile(L.bias, 1, h), c2 = bias_tile(L.bias, 2, h),
               c3 = bias_tile(L.bias, 3, h);
        mac4(L.w4 + lane, L.k8, bptr, c0, c1, c2, c3);
        const unsigned m0 = relu_tile(act, 0, j, h, c0);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned m1 = relu_tile(act, 1, j, h, c1);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned m2 = relu_tile(act, 2, j, h, c2);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned m3 = relu_tile(act, 3, j, h, c3);
        a.mask[l][tile * 64 + lane] = (unsigned long long)(m0 | (m1 << 16)) | ((unsigned long long)(m2 | (m3 << 16)) << 32);
    }
    // classifier head (13 of 32 rows real) + log_softmax
    {
        f32x16 acc = bias_tile(a.head.bias, 0, h);
        acc = tile_mac<WBLK, false>(a.head.w + lane, a.head.k8, bptr, acc);
        wave_store_tile(act, 0, j, h, acc);
    }
    for (int t = lane; t < 32 * 8; t += 64) {
        const int p = t >> 3, q = t & 7;
        const bool has0 = q < a.n_cls, has1 = q + 8 < a.n_cls;
        const float z0 = has0 ? act[p * 8 + q] : -INFINITY;
        const float z1 = has1 ? act[WBLK + p * 8 + q] : -INFINITY;
        float m = fmaxf(z0, z1);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        m = fmaxf(m, __shfl_xor(m, 4));
        float s = (has0 ? expf(z0 - m) : 0.0f) + (has1 ? expf(z1 - m) : 0.0f);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        const float lse = logf(s);
        float *o = a.logp + ((size_t)b * a.N + n0 + p) * a.n_cls;
        if (has0) o[q] = (z0 - m) - lse;
        if (has1) o[q + 8] = (z1 - m) - lse;
    }
}

// ------------------------------------------------------------------------------------ fp1 + head, backward
struct Fp1BwdArgs {
    const float *logp;        // [B][N][n_cls]
    const float *dlogp;       // [B][N][n_cls]
    const unsigned long long *mask[4];   // as written by the forward kernel (layer order mlp0, mlp1, mlp2, conv1)
    ChainLayer layer[5];      // conv2^T (k8 = 4), conv1^T, mlp2^T, mlp1^T, mlp0^T  (bias = null)
    float *dint_out;          // [B][N][128] gradient of the interpolated features, plain rows
    int N, n_cls;
};

__global__ __launch_bounds__(64) void fp1_bwd_wave_kernel(Fp1BwdArgs a)
{
    extern __shared__ float act[];
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    int bx, b;
    xcd_tile(bx, b);
    const int n0 = bx * 32;
    const size_t tile = (size_t)b * gridDim.x + bx;

    // log_softmax backward: dz = dlogp - softmax * sum(dlogp)   (rows 0..n_cls-1 of a 32-row, zero padded block)
    if (lane < 32) {
        const size_t n = (size_t)b * a.N + n0 + lane;
        const float *dl = a.dlogp + n * a.n_cls, *lp = a.logp + n * a.n_cls;
        float s = 0.0f;
        for (int c = 0; c < a.n_cls; ++c) s += dl[c];
        float z[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) z[c] = c < a.n_cls ? dl[c] - expf(lp[c]) * s : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4 *)(act + (size_t)(q >> 1) * WBLK + lane * 8 + 4 * (q & 1)) = make_float4(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]);
    } else {
        const int p = lane - 32;   // K-padding blocks 2, 3 of conv2^T
#pragma unroll
        for (int blk = 2; blk < 4; ++blk) {
            float *zp = act + (size_t)blk * WBLK + p * 8;
            *(float4 *)zp = make_float4(0.f, 0.f, 0.f, 0.f);
            *(float4 *)(zp + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float *bptr = act + j * 8 + 4 * h;
#pragma unroll 1
    for (int l = 0; l < 5; ++l) {
        const ChainLayer L = a.layer[l];
        // ReLU mask of the activation this layer's output is the gradient of: conv1, mlp2, mlp1, mlp0, none
        unsigned long long m = ~0ull;
        if (l < 4) m = a.mask[3 - l][tile * 64 + lane];
        f32x16 c0 = bias_tile(nullptr, 0, h), c1 = c0, c2 = c0, c3 = c0;
        mac4(L.w4 + lane, L.k8, bptr, c0, c1, c2, c3);
        mask_tile(act, 0, j, h, c0, (unsigned)m & 0xFFFFu);
        __builtin_amdgcn_sched_barrier(0);
        mask_tile(act, 1, j, h, c1, (unsigned)(m >> 16) & 0xFFFFu);
        __builtin_amdgcn_sched_barrier(0);
        mask_tile(act, 2, j, h, c2, (unsigned)(m >> 32) & 0xFFFFu);
        __builtin_amdgcn_sched_barrier(0);
        mask_tile(act, 3, j, h, c3, (unsigned)(m >> 48) & 0xFFFFu);
    }
    // gradient rows of the interpolated features: 32 lanes x float4 = one 512-byte row, two rows per pass
#pragma unroll 4
    for (int p = 0; p < 32; p += 2) {
        const int pp = p + h;
        const float4 v = *(const float4 *)(act + (size_t)(j >> 1) * WBLK + pp * 8 + 4 * (j & 1));
        *(float4 *)(a.dint_out + ((size_t)b * a.N + n0 + pp) * 128 + 4 * j) = v;
    }
}

}  // namespace psg
