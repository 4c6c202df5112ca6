// Device building blocks of the fused per-point MLP chains (shared 1x1-conv stacks) on gfx950.
//
// Activations of one workgroup's P points live in LDS as act[channel][point] (leading dimension
// LDP = P+1 floats).  A layer is a sequence of 32x32 output tiles, one MFMA accumulator each:
//     D[m = out channel][n = point] += A[m][k] * B[k][n]       v_mfma_f32_32x32x2_f32
// A = weights, pre-packed on the host so that a wave reads one coalesced float4 per lane per four
// MFMAs; B = the activation tile, one conflict-free ds_read_b32 per MFMA.  fp32 in / fp32
// accumulate: the matrix pipe computes an exact fmaf chain, so results match a CPU fp32 GEMM to
// rounding (no reduced-precision path exists or is wanted on this path).
//
// k-ordering inside an 8-wide k chunk: MFMA step t (0..3) consumes k = 8*k8 + 2*t + h from lane
// half h, for A and B alike (any consistent k permutation is a valid dot product).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int MAX_LAYERS = 5;

struct FwdLayer {
    const float4 *w;    // packed [mb][k8][64] float4: lane (i,h) elem t = W[mb*32+i][8*k8+2*t+h]
    const float *bias;  // [mb*32], zero padded
    uint16_t *mask;     // ReLU mask out, [wg][mb][pb][64] (bit r of a lane = its accumulator r > 0), or null
    int k8, mb, relu;
};

struct BwdLayer {
    const float4 *w;       // packed transpose: lane (i,h) elem t = W[8*k8+2*t+h][mb*32+i]
    const uint16_t *mask;  // mask of the activation this layer's OUTPUT is the gradient of, or null
    int k8, mb;
};

// accumulator register r of lane half h holds output row (r&3) + 8*(r>>2) + 4*h of the 32x32 tile
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// One 32x32 output tile: acc += sum_k A[.,k] B[k,.].  Weights stream from global/L2 with PF packed
// float4 chunks in flight per lane (an L2 hit costs ~500 cycles, one chunk feeds 4 MFMAs = 256 cycles
// of matrix pipe, so PF = 4 covers the latency with one wave and leaves slack with two per SIMD).
// FLIP swaps the MFMA operands: D[point][channel] instead of D[channel][point].
template <int LDP, bool FLIP>
__device__ __forceinline__ f32x16 tile_mac(const float4 *__restrict__ w, int k8n, const float *__restrict__ bptr,
                                           f32x16 acc)
{
    // w already offset to [mb][0][lane]; bptr = act + h*LDP + pb*32 + (lane&31)
    constexpr int PF = 4;
    float4 buf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) buf[i] = w[(size_t)min(i, k8n - 1) * 64];
    for (int k8 = 0; k8 < k8n; k8 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (k8 + i < k8n) {
                const float4 a = buf[i];
                buf[i] = w[(size_t)min(k8 + i + PF, k8n - 1) * 64];  // tail re-reads the last chunk (unused)
                const float *bp = bptr + (size_t)(k8 + i) * 8 * LDP;
                const float b0 = bp[0], b1 = bp[2 * LDP], b2 = bp[4 * LDP], b3 = bp[6 * LDP];
                if (FLIP) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b2, a.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b3, a.w, acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b3, acc, 0, 0, 0);
                }
            }
        }
    }
    return acc;
}

// Forward layer: out[m][p] = act(W in[:, p] + b).  Tiles (mb x P/32) are dealt round-robin to waves.
template <int P, int NW>
__device__ __forceinline__ void layer_fwd(const FwdLayer &L, const float *__restrict__ in, float *__restrict__ out,
                                          size_t wg_linear)
{
    constexpr int LDP = P + 1, PB = P / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntask = L.mb * PB;
    for (int task = wave; task < ntask; task += NW) {
        const int mb = task / PB, pb = task - mb * PB;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = L.bias[mb * 32 + acc_row(r, h)];
        acc = tile_mac<LDP, false>(L.w + (size_t)mb * L.k8 * 64 + lane, L.k8, in + h * LDP + pb * 32 + j, acc);
        unsigned m = 0;
        float *o = out + (size_t)(mb * 32) * LDP + pb * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            if (L.relu) {
                bool pos = v > 0.0f;
                v = pos ? v : 0.0f;
                m |= (unsigned)pos << r;
            }
            o[acc_row(r, h) * LDP] = v;
        }
        if (L.mask) L.mask[(wg_linear * ntask + task) * 64 + lane] = (uint16_t)m;
    }
}

// Backward (input-gradient) layer: out[m][p] = mask * (W^T in[:, p]).
template <int P, int NW>
__device__ __forceinline__ void layer_bwd(const BwdLayer &L, const float *__restrict__ in, float *__restrict__ out,
                                          size_t wg_linear)
{
    constexpr int LDP = P + 1, PB = P / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntask = L.mb * PB;
    for (int task = wave; task < ntask; task += NW) {
        const int mb = task / PB, pb = task - mb * PB;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = tile_mac<LDP, false>(L.w + (size_t)mb * L.k8 * 64 + lane, L.k8, in + h * LDP + pb * 32 + j, acc);
        unsigned m = 0xFFFFu;
        if (L.mask) m = L.mask[(wg_linear * ntask + task) * 64 + lane];
        float *o = out + (size_t)(mb * 32) * LDP + pb * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[acc_row(r, h) * LDP] = ((m >> r) & 1u) ? acc[r] : 0.0f;
    }
}

}  // namespace psg
