// Wave-level helpers shared by the kNN translation units (psg_resgcn.hip, psg_knn.hip): DPP reductions over the 64 lanes,
// the order-preserving float <-> unsigned key map and the in-register bitonic sort of composite keys.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_get(unsigned v)   // lanes of rows outside ROW_MASK read 0
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// sum over the 64 lanes, wave-uniform result (row butterflies, then row_bcast15 / row_bcast31 into lane 63)
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
    v += dpp_get<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xF>(v);   // row_half_mirror
    v += dpp_get<0x140, 0xF>(v);   // row_mirror: every lane holds its row's sum
    v += dpp_get<0x142, 0xA>(v);   // row_bcast15 -> rows 1, 3
    v += dpp_get<0x143, 0xC>(v);   // row_bcast31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false); v = o > v ? o : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned key_of(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dist_of(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// Bitonic sort of M = 64 * SL composite keys held SL per lane (element e = lane * SL + slot): strides below SL
// are compare-exchanges between a lane's own registers, strides of SL and more exchange whole registers with the
// partner lane (two 32-bit cross-lane permutes per key); nothing goes through LDS memory and there is no fence per
// step (the LDS version spent 23 us of the 60 us selection here).
template <int SL, int STRIDE>
__device__ __forceinline__ void sort_intra(unsigned long long (&v)[SL], int lane, unsigned size)
{
    if constexpr (STRIDE >= 1 && STRIDE < SL) {
#pragma unroll
        for (int s0 = 0; s0 < SL; ++s0) {
            if ((s0 & STRIDE) == 0) {
                const unsigned e = (unsigned)lane * SL + s0;
                const bool up = (e & size) == 0;
                const unsigned long long a = v[s0], b = v[s0 | STRIDE];
                const bool sw = (a > b) == up;
                v[s0] = sw ? b : a;
                v[s0 | STRIDE] = sw ? a : b;
            }
        }
    }
}

template <int SL>
__device__ __forceinline__ void wave_sort_keys(unsigned long long *cand, int lane)
{
    unsigned long long v[SL];
#pragma unroll
    for (int s0 = 0; s0 < SL; ++s0) v[s0] = cand[lane * SL + s0];
    constexpr unsigned M = 64u * SL;
    for (unsigned size = 2; size <= M; size <<= 1) {
        for (unsigned stride = size >> 1; stride >= (unsigned)SL; stride >>= 1) {   // partner lane = lane ^ (stride / SL)
            const int ls = (int)(stride / SL);
            const bool lower = (lane & ls) == 0;
#pragma unroll
            for (int s0 = 0; s0 < SL; ++s0) {
                const unsigned e = (unsigned)lane * SL + s0;
                const bool up = (e & size) == 0;
                const unsigned long long o =
                    ((unsigned long long)__shfl_xor((unsigned)(v[s0] >> 32), ls) << 32) | __shfl_xor((unsigned)v[s0], ls);
                const bool keep_min = lower == up;
                v[s0] = keep_min ? (o < v[s0] ? o : v[s0]) : (o > v[s0] ? o : v[s0]);
            }
        }
        if (size > 4) sort_intra<SL, 4>(v, lane, size);
        if (size > 2) sort_intra<SL, 2>(v, lane, size);
        sort_intra<SL, 1>(v, lane, size);
    }
#pragma unroll
    for (int s0 = 0; s0 < SL; ++s0) cand[lane * SL + s0] = v[s0];
}

}  // namespace
