// Producer-side helpers of the fused feature-space kNN (C = 64): whoever writes a block's 64 output features of a vertex
// (edge_max_fwd_kernel in the network, knn_prep_kernel for the stand-alone entry) also writes the copies the kNN kernels
// stream, one thread per (vertex v, channel c), 64 consecutive lanes = one vertex:
//   xp  fp32, v_mfma_f32_16x16x4_f32 operand order  [v / 16][4][64 lanes] float4           (psg_knn_fused.cuh: exact path)
//   bp  bf16 hi / lo split, v_mfma_f32_32x32x16_bf16 operand order  [v / 32][9][64 lanes] 8 x bf16   (psg_knn_bf.cuh):
//       k-steps 0..3 = bf16(x), 4..7 = bf16(x - bf16(x)), 8 = the augmented step that folds -|x|^2 / 2 into the product
//   sq  |x|^2 in torch.sum's order (ResGCN/gcn_lib/dense/torch_edge.py:42, SURVEY 8a')
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// the per-pair error term of the bf16 prefilter (psg_knn_bf.cuh): |D~ - D| <= A |x_i||x_j| + B |x_j|^2 + G |x_i|^2
// (round 6: 2^-13.5 / 2^-15.5 -> 2^-14 / 2^-19 with the small products accumulated first, derivation in psg_knn_bf.cuh)
#ifndef KNN_BF_A_VALUE
#define KNN_BF_A_VALUE 6.103515625e-05f               // 2^-14
#define KNN_BF_B_VALUE 1.9073486328125e-06f           // 2^-19
#endif
constexpr float KNN_BF_A = KNN_BF_A_VALUE;
constexpr float KNN_BF_B = KNN_BF_B_VALUE;
constexpr float KNN_BF_G = 9.5367431640625e-07f;      // 2^-20

// round-to-nearest-even bf16 of a finite float (non-finite inputs never reach the bf16 path: the kernel falls back)
__device__ __forceinline__ unsigned short knn_bf16_rne(float f)
{
    const unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float knn_bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ void knn_store_xp(float *__restrict__ xp, size_t v, int c, float val)
{
    const int s4 = c >> 2, g4 = c & 3;
    xp[(v >> 4) * 1024 + (size_t)((((s4 >> 2) * 64) + (int)(v & 15) + 16 * g4) * 4 + (s4 & 3))] = val;
}

// fragment of k-step s (16 features), lane (h = (k >> 3) & 1, r = v & 31), element k & 7
__device__ __forceinline__ void knn_store_bp(unsigned short *__restrict__ bp, size_t v, int c, float val)
{
    const unsigned short hi = knn_bf16_rne(val);
    const unsigned short lo = knn_bf16_rne(val - knn_bf16_f32(hi));       // (the difference is exact in fp32)
    const size_t frag = ((v >> 5) * 9 + (size_t)(c >> 4)) * 64 + (size_t)(((c >> 3) & 1) * 32 + (int)(v & 31));
    bp[frag * 8 + (c & 7)] = hi;
    bp[(frag + 4 * 64) * 8 + (c & 7)] = lo;
}

// augmented k-step of vertex v, elements 0..3 of the h = 0 lane (zeros elsewhere): -(1 - B) sq / 2 as three bf16 pieces (24
// significant bits), multiplied by 1, 1, 1 on the query side, and (A / 2) |x| rounded UP to bf16, multiplied by the query's
// |x| rounded up: the product term folds the pair's error bound into the accumulator (psg_knn_bf.cuh).  One thread per vertex.
__device__ __forceinline__ void knn_store_aug(uint4 *__restrict__ bp4, size_t v, float sq)
{
    const float hj = -0.5f * (1.0f - KNN_BF_B) * sq;
    const unsigned short p1 = knn_bf16_rne(hj);
    const float r1 = hj - knn_bf16_f32(p1);
    const unsigned short p2 = knn_bf16_rne(r1);
    const unsigned short p3 = knn_bf16_rne(r1 - knn_bf16_f32(p2));
    const unsigned short p4 = (unsigned short)((__float_as_uint(0.5f * KNN_BF_A * sqrtf(sq)) + 0xFFFFu) >> 16);
    const size_t frag = ((v >> 5) * 9 + 8) * 64 + (size_t)(v & 31);
    bp4[frag] = make_uint4((unsigned)p1 | ((unsigned)p2 << 16), (unsigned)p3 | ((unsigned)p4 << 16), 0u, 0u);
    bp4[frag + 32] = make_uint4(0u, 0u, 0u, 0u);
}

// torch.sum(x * x, -1) for the 64 contiguous floats held one per lane (lane = channel): u_c = x_c^2 + x_{c+32}^2,
// t_l = ((u_l + u_{8+l}) + u_{16+l}) + u_{24+l}, s = t_0 + t_1 + ... + t_7 left to right; the result is valid in every lane.
// (compile with -ffp-contract=off: hipcc's __fmul_rn / __fadd_rn are plain operators)
__device__ __forceinline__ float knn_wave_sumsq(float val, int lane)
{
    const float q2 = __fmul_rn(val, val);
    const float u = __fadd_rn(q2, __shfl(q2, (lane + 32) & 63));
    float tl = __fadd_rn(u, __shfl(u, (lane + 8) & 63));
    tl = __fadd_rn(tl, __shfl(u, (lane + 16) & 63));
    tl = __fadd_rn(tl, __shfl(u, (lane + 24) & 63));      // valid in lanes 0..7
    float sacc = __shfl(tl, 0);
#pragma unroll
    for (int l = 1; l < 8; ++l) sacc = __fadd_rn(sacc, __shfl(tl, l));
    return sacc;
}

}  // namespace
