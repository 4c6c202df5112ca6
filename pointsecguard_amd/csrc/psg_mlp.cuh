// Device building blocks of the fused per-point MLP chains (shared 1x1-conv stacks) on gfx950.
//
// Activations of one workgroup's P points live in LDS in "k8-block" layout
//     act[c / 8][point][c % 8]        (block stride BLK = P*8 floats)
// so that BOTH sides of a layer are 16-byte accesses that are contiguous across the wave:
//   * MFMA operand read: lane (j = lane&31, h = lane>>5) reads the float4 at [k8][pb*32+j][4h..4h+3]
//     (one ds_read_b128 feeds 4 MFMAs; the 64 lanes cover 1 KiB contiguous -> conflict free);
//   * accumulator write-back: registers 4g..4g+3 of lane (j,h) are channels mb*32 + 8g + 4h + (0..3)
//     of point j, i.e. one ds_write_b128 at [mb*4+g][pb*32+j][4h] (again 1 KiB contiguous per wave).
//
// A layer is a sequence of 32x32 output tiles, one MFMA accumulator each:
//     D[m = out channel][n = point] += A[m][k] * B[k][n]       v_mfma_f32_32x32x2_f32
// A = weights, pre-packed on the host so that a wave reads one coalesced float4 per lane per four
// MFMAs (k = 8*k8 + 4*h + t for MFMA step t, matching the activation read above; any consistent k
// permutation is a valid dot product).  fp32 in / fp32 accumulate: the matrix pipe computes an exact
// fmaf chain, so results match a CPU fp32 GEMM to rounding (no reduced-precision path on this path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// LDS block padding / spare blocks.  Both default to 0: the MFMA operand reads and the tile write-backs are
// 1 KiB-contiguous per wave whatever the block stride, and the occupancy the smaller footprint buys (sa1 backward:
// 2 -> 3 workgroups per CU, sa2: 3 -> 4) is worth more than the bank conflicts of the row-wise epilogue reads that
// the +8 padding used to avoid (measured: 471 -> 481 rooms/s).
#ifndef PSG_LDS_PAD
#define PSG_LDS_PAD 0
#endif
#ifndef PSG_LDS_SPARE
#define PSG_LDS_SPARE 0
#endif

namespace psg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int MAX_LAYERS = 5;

struct FwdLayer {
    const float4 *w;    // packed [mb][k8][64] float4: lane (i,h) elem t = W[mb*32+i][col(8*k8+4*h+t)]
    const float *bias;  // [mb*32], zero padded
    uint16_t *mask;     // ReLU mask out, [wg][mb][pb][64] (bit r of a lane = its accumulator r > 0), or null
    int k8, mb, relu;
};

struct BwdLayer {
    const float4 *w;       // packed transpose: lane (i,h) elem t = W[8*k8+4*h+t][col(mb*32+i)]
    const uint16_t *mask;  // mask of the activation this layer's OUTPUT is the gradient of, or null
    int k8, mb;
};

template <int P> struct Lds {
    static constexpr int BLK = P * 8 + PSG_LDS_PAD;  // floats per 8-channel block
    __device__ static __forceinline__ int off(int c, int p) { return (c >> 3) * BLK + p * 8 + (c & 7); }
};

// accumulator register r of lane half h holds output row (r&3) + 8*(r>>2) + 4*h of the 32x32 tile
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <bool FLIP>
__device__ __forceinline__ f32x16 mfma4(const float4 a, const float4 b, f32x16 acc)
{
    if (FLIP) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    return acc;
}

// One 32x32 output tile: acc += sum_k A[.,k] B[k,.] over k8n chunks, k8n a positive multiple of 4 (see tile_mac
// below for the general case).
//
// The k-loop is hand-scheduled in inline assembly.  hipcc cannot express this pipeline: left to itself it
// sinks every weight re-load to just before its use, and with the order pinned by sched_barrier it still
// drains vmcnt(0) at the loop head (its wait-count pass does not carry counted loads across the back
// edge).  Steady state per iteration (16 MFMAs = 1024 matrix-pipe cycles):
//   * four weight chunks (global_load_dwordx4, 1 KiB per wave) live in a static ring v[A0..A3]; a chunk
//     is re-loaded right after its last use and waited for with a COUNTED s_waitcnt vmcnt(3), i.e. it
//     has three younger loads and ~1000 cycles of MFMA behind it when it is needed;
//   * the activation chunk (ds_read_b128) is fetched one step ahead (lgkmcnt(1)).
// The last four chunks run in a peeled pass without re-loads, so nothing is fetched past the tile.  Fixed VGPRs v[64:87] are used for the operand ring and
// declared as clobbers (keeps the kernels at <= 128 VGPRs = 4 waves per SIMD).
// FLIP swaps the MFMA operands: D[point][channel] instead of D[channel][point].
#if defined(PSG_DIAG_BUILD) && defined(PSG_DIAG_NOLOAD)  // timing experiment only: the k-loop re-uses the first four weight chunks
#define PSG_RELOAD(x) ""
#else
#define PSG_RELOAD(x) x
#endif
template <int BLK, bool FLIP>
__device__ __forceinline__ f32x16 tile_mac4(const float4 *__restrict__ w, int k8n, const float *__restrict__ bptr,
                                            f32x16 acc)
{
    // w already offset to [mb][0][lane]; bptr = act + (pb*32 + (lane&31))*8 + 4*(lane>>5)
    unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)bptr;
    const float4 *wp = w;
    // pipelined iterations; the last 4 chunks run without re-loads.  A wave-uniform count that the k-loop wants in an SGPR - and
    // that the compiler does not always keep there: behind layer_bwd's per-wave task selection (fp_bwd_kernel<32, 8, 2> with
    // the FP-split layers) it lands in a VGPR and the build ends in "illegal VGPR to SGPR copy", while
    // __builtin_amdgcn_readfirstlane is folded away where the value is provably uniform.  So the move is spelled out; the
    // s_nop covers gfx940+'s one wait state between a VALU write of a VGPR and a v_readfirstlane of it, which the hazard
    // recognizer does not insert inside inline assembly (without it the count was read stale: a memory fault).
    // Round 6 (advisor): the value is laundered through an EMPTY asm (the compiler can no longer prove it uniform and fold the
    // builtin away) and moved by the compiler's own v_readfirstlane, so that the hazard recognizer sees both sides: the VGPR
    // write in front of it and whatever reads the SGPR behind it (tools/check_asm_hazards.py lints the boundary cases).
    int n_v = (k8n >> 2) - 1;
    asm volatile("" : "+v"(n_v));
    int n = __builtin_amdgcn_readfirstlane(n_v);
    const unsigned long long step = 4096ull;  // 4 chunks x 64 lanes x 16 B
    constexpr int S1 = BLK * 4, S2 = 2 * BLK * 4, S3 = 3 * BLK * 4, S4 = 4 * BLK * 4;
    if (FLIP) {
        asm volatile(
            "global_load_dwordx4 v[64:67], %[wp], off\n\t"
            "global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t"
            "global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t"
            "global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t"
            "ds_read_b128 v[80:83], %[lds]\n\t"
            "s_cmp_eq_u32 %[n], 0\n\t"
            "s_cbranch_scc1 L_psg_last_%=\n\t"
            "L_psg_loop_%=:\n\t"
            "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
            "ds_read_b128 v[84:87], %[lds] offset:%[s1]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v80, v64, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v81, v65, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v82, v66, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v83, v67, %[acc]\n\t"
            PSG_RELOAD("global_load_dwordx4 v[64:67], %[wp], off\n\t")
            "ds_read_b128 v[80:83], %[lds] offset:%[s2]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v84, v68, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v85, v69, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v86, v70, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v87, v71, %[acc]\n\t"
            PSG_RELOAD("global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t")
            "ds_read_b128 v[84:87], %[lds] offset:%[s3]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v80, v72, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v81, v73, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v82, v74, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v83, v75, %[acc]\n\t"
            PSG_RELOAD("global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t")
            "ds_read_b128 v[80:83], %[lds] offset:%[s4]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v84, v76, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v85, v77, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v86, v78, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v87, v79, %[acc]\n\t"
            PSG_RELOAD("global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t")
            "v_add_u32 %[lds], %[s4], %[lds]\n\t"
            "s_sub_u32 %[n], %[n], 1\n\t"
            "s_cmp_lg_u32 %[n], 0\n\t"
            "s_cbranch_scc1 L_psg_loop_%=\n\t"
            "L_psg_last_%=:\n\t"
            "ds_read_b128 v[84:87], %[lds] offset:%[s1]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v80, v64, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v81, v65, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v82, v66, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v83, v67, %[acc]\n\t"
            "ds_read_b128 v[80:83], %[lds] offset:%[s2]\n\t"
            "s_waitcnt vmcnt(2) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v84, v68, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v85, v69, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v86, v70, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v87, v71, %[acc]\n\t"
            "ds_read_b128 v[84:87], %[lds] offset:%[s3]\n\t"
            "s_waitcnt vmcnt(1) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v80, v72, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v81, v73, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v82, v74, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v83, v75, %[acc]\n\t"
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v84, v76, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v85, v77, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v86, v78, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v87, v79, %[acc]\n\t"
            "s_nop 15\n\t"
            "s_nop 3\n\t"
            : [acc] "+v"(acc), [wp] "+v"(wp), [lds] "+v"(lds), [n] "+s"(n)
            : [step] "s"(step), [s1] "n"(S1), [s2] "n"(S2), [s3] "n"(S3), [s4] "n"(S4)
            : "memory", "scc", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
              "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    } else {
        asm volatile(
            "global_load_dwordx4 v[64:67], %[wp], off\n\t"
            "global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t"
            "global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t"
            "global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t"
            "ds_read_b128 v[80:83], %[lds]\n\t"
            "s_cmp_eq_u32 %[n], 0\n\t"
            "s_cbranch_scc1 L_psg_last_%=\n\t"
            "L_psg_loop_%=:\n\t"
            "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
            "ds_read_b128 v[84:87], %[lds] offset:%[s1]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v64, v80, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v65, v81, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v66, v82, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v67, v83, %[acc]\n\t"
            PSG_RELOAD("global_load_dwordx4 v[64:67], %[wp], off\n\t")
            "ds_read_b128 v[80:83], %[lds] offset:%[s2]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v68, v84, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v69, v85, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v70, v86, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v71, v87, %[acc]\n\t"
            PSG_RELOAD("global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t")
            "ds_read_b128 v[84:87], %[lds] offset:%[s3]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v72, v80, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v73, v81, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v74, v82, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v75, v83, %[acc]\n\t"
            PSG_RELOAD("global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t")
            "ds_read_b128 v[80:83], %[lds] offset:%[s4]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v76, v84, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v77, v85, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v78, v86, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v79, v87, %[acc]\n\t"
            PSG_RELOAD("global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t")
            "v_add_u32 %[lds], %[s4], %[lds]\n\t"
            "s_sub_u32 %[n], %[n], 1\n\t"
            "s_cmp_lg_u32 %[n], 0\n\t"
            "s_cbranch_scc1 L_psg_loop_%=\n\t"
            "L_psg_last_%=:\n\t"
            "ds_read_b128 v[84:87], %[lds] offset:%[s1]\n\t"
            "s_waitcnt vmcnt(3) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v64, v80, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v65, v81, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v66, v82, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v67, v83, %[acc]\n\t"
            "ds_read_b128 v[80:83], %[lds] offset:%[s2]\n\t"
            "s_waitcnt vmcnt(2) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v68, v84, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v69, v85, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v70, v86, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v71, v87, %[acc]\n\t"
            "ds_read_b128 v[84:87], %[lds] offset:%[s3]\n\t"
            "s_waitcnt vmcnt(1) lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v72, v80, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v73, v81, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v74, v82, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v75, v83, %[acc]\n\t"
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v76, v84, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v77, v85, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v78, v86, %[acc]\n\t"
            "v_mfma_f32_32x32x2_f32 %[acc], v79, v87, %[acc]\n\t"
            "s_nop 15\n\t"
            "s_nop 3\n\t"
            : [acc] "+v"(acc), [wp] "+v"(wp), [lds] "+v"(lds), [n] "+s"(n)
            : [step] "s"(step), [s1] "n"(S1), [s2] "n"(S2), [s3] "n"(S3), [s4] "n"(S4)
            : "memory", "scc", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
              "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    }
    return acc;
}

// The 18 wait states between a 16-pass MFMA and the first vector instruction that touches its result, spelled out for
// the places where the COMPILER's MFMAs (the K tails below) are followed by a consumer inside inline assembly (relu_bits):
// the hazard recognizer counts wait states only between instructions it can see, so it emitted v_mfma x 4, v_cmp_lt_f32 on
// the accumulator back to back (tools/check_asm_hazards.py, round 6: 36 sites in the sa_fwd kernels).  Results were right
// all the same - every K tail of this network ends in a zero-padded k-step, and a stale read sees the accumulator one
// dependent MFMA earlier - but that is luck, not design.
__device__ __forceinline__ void mfma_fence(f32x16 &acc) { asm volatile("s_nop 15\n\ts_nop 1" : "+v"(acc)); }

// ReLU of an accumulator tile in place + its 16 "was positive" bits (bit r = register r).  Three vector instructions
// per register (compare, select, add-with-carry shifts the bit in: registers 15 .. 0 so bit r lands at position r); the
// compiler's version of `pos ? c : 0; m |= pos << r` took five to six, and on this chip a vector instruction of an
// epilogue costs the matrix pipe its issue slot one for one (tools/mfma_valu_overlap.hip).
__device__ __forceinline__ unsigned relu_bits(f32x16 &c)
{
    unsigned m = 0;
#define PSG_RB(r)                                                                                                          \
    {                                                                                                                      \
        float x = c[r];                                                                                                    \
        asm("v_cmp_lt_f32 vcc, 0, %0\n\tv_cndmask_b32 %0, 0, %0, vcc\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"              \
            : "+v"(x), "+v"(m) : : "vcc");                                                                                 \
        c[r] = x;                                                                                                          \
    }
    PSG_RB(15) PSG_RB(14) PSG_RB(13) PSG_RB(12) PSG_RB(11) PSG_RB(10) PSG_RB(9) PSG_RB(8)
    PSG_RB(7) PSG_RB(6) PSG_RB(5) PSG_RB(4) PSG_RB(3) PSG_RB(2) PSG_RB(1) PSG_RB(0)
#undef PSG_RB
    return m;
}

// Gradient mask: c[r] = bit r of m ? c[r] : 0, as an AND with the sign-extended bit (two instructions per register).
__device__ __forceinline__ void apply_bits(f32x16 &c, unsigned m)
{
#pragma unroll
    for (int r = 0; r < 16; ++r)
        c[r] = __uint_as_float(__float_as_uint(c[r]) & (unsigned)__builtin_amdgcn_sbfe((int)m, r, 1));
}

// One 32x32 output tile over k8n chunks of 8 input channels: the multiple-of-4 part runs in the pipelined assembly
// loop above, a tail of 1..3 chunks (first layers whose K is not a multiple of 32: 12+4, 67+5, 131+5, 259+5 channels,
// and the 13-class head's transpose) as plain MFMAs, instead of padding K to 32 with zero work.
template <int BLK, bool FLIP>
__device__ __forceinline__ f32x16 tile_mac(const float4 *__restrict__ w, int k8n, const float *__restrict__ bptr,
                                           f32x16 acc)
{
    const int k4 = k8n & ~3;
    if (k4) acc = tile_mac4<BLK, FLIP>(w, k4, bptr, acc);
    for (int k8 = k4; k8 < k8n; ++k8) {
        const float4 a = w[(size_t)k8 * 64];
        const float4 b = *(const float4 *)(bptr + (size_t)k8 * BLK);
        acc = mfma4<FLIP>(a, b, acc);
    }
    if (k8n > k4) mfma_fence(acc);
    return acc;
}

// write a D[channel][point] accumulator tile back to LDS: 4 x ds_write_b128 per lane
template <int P>
__device__ __forceinline__ void store_tile(float *__restrict__ out, int mb, int pcol, int h, const f32x16 &v)
{
    float *o = out + (size_t)(mb * 4) * Lds<P>::BLK + pcol * 8 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *(float4 *)(o + (size_t)g * Lds<P>::BLK) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

// Layers run IN PLACE on one LDS buffer: every wave first computes its tiles into registers (up to MAXT tiles of
// a ragged layer), the workgroup meets at a barrier (all reads of the input are complete), then the tiles are
// written back over the input.  One buffer of max(K, M) channels instead of an input and an output buffer: the
// LDS footprint, not registers, is what bounds the number of co-resident workgroups of these kernels, and more of
// them is what hides a workgroup's gather / epilogue / barrier phases (sa2 backward: 2 -> 4 workgroups per CU).
// The caller places a barrier after the layer, as before.
//
// Tiles (mb x P/32) are dealt round-robin to waves, starting at a wave that rotates with the workgroup: a layer
// with fewer tiles than waves (the 13-class head: one tile) or a ragged count (10 tiles on 8 waves) would otherwise
// always load the same SIMDs of the CU.

// Forward layer: buf[m][p] <- act(W buf[:, p] + b).
template <int P, int NW, int MAXT>
__device__ __forceinline__ void layer_fwd(const FwdLayer &L, float *__restrict__ buf, size_t wg_linear)
{
    constexpr int PB = P / 32, BLK = Lds<P>::BLK;
    static_assert((NW & (NW - 1)) == 0, "NW must be a power of two");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntask = L.mb * PB;
    const int first = (wave + (int)(wg_linear & (NW - 1))) & (NW - 1);
    f32x16 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int task = first + i * NW;
        if (task < ntask) {
            const int mb = task / PB, pb = task - mb * PB;
            // bias: 4 x float4 issued before the k-loop and consumed after it, so the load latency hides behind the
            // MFMAs (initialising the accumulators from it - no zeroing moves, no adds - was measured: no faster, and the
            // changed summation order moved one MSG oracle comparison past its tolerance)
            const float4 *bp = (const float4 *)(L.bias + mb * 32 + 4 * h);
            const float4 bq0 = bp[0], bq1 = bp[2], bq2 = bp[4], bq3 = bp[6];
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.0f;
            c = tile_mac<BLK, false>(L.w + (size_t)mb * L.k8 * 64 + lane, L.k8, buf + (pb * 32 + j) * 8 + 4 * h, c);
            c[0] += bq0.x; c[1] += bq0.y; c[2] += bq0.z; c[3] += bq0.w;
            c[4] += bq1.x; c[5] += bq1.y; c[6] += bq1.z; c[7] += bq1.w;
            c[8] += bq2.x; c[9] += bq2.y; c[10] += bq2.z; c[11] += bq2.w;
            c[12] += bq3.x; c[13] += bq3.y; c[14] += bq3.z; c[15] += bq3.w;
            if (L.relu) {
                const unsigned m = relu_bits(c);
                if (L.mask) L.mask[(wg_linear * ntask + task) * 64 + lane] = (uint16_t)m;
            }
            acc[i] = c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int task = first + i * NW;
        if (task < ntask) {
            const int mb = task / PB, pb = task - mb * PB;
            store_tile<P>(buf, mb, pb * 32 + j, h, acc[i]);
        }
    }
}

// accumulate a tile into LDS: buf[tile] += v  (same addressing as store_tile)
template <int P>
__device__ __forceinline__ void add_tile(float *__restrict__ out, int mb, int pcol, int h, const f32x16 &v)
{
    float *o = out + (size_t)(mb * 4) * Lds<P>::BLK + pcol * 8 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 c = *(float4 *)(o + (size_t)g * Lds<P>::BLK);
        c.x += v[4 * g]; c.y += v[4 * g + 1]; c.z += v[4 * g + 2]; c.w += v[4 * g + 3];
        *(float4 *)(o + (size_t)g * Lds<P>::BLK) = c;
    }
}

// Backward (input-gradient) layer: buf[m][p] <- mask * (W^T buf[:, p]).
//
// Ragged layers (10 tiles on 8 waves, 5 on 4, 9 on 8 ...) would leave most waves idle in their last round: the
// `rem` left-over tiles are instead split along K over groups of gsz = min(NW / rem, 4) waves (every wave gets a
// K / gsz slice of one left-over tile), and the partial tiles are summed in LDS in slice order, one barrier per
// slice (a fixed order: the result stays bit-reproducible; LDS float atomics would not be).
template <int P, int NW, int MAXT>
__device__ __forceinline__ void layer_bwd(const BwdLayer &L, float *__restrict__ buf, size_t wg_linear)
{
    constexpr int PB = P / 32, BLK = Lds<P>::BLK;
    static_assert((NW & (NW - 1)) == 0, "NW must be a power of two");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntask = L.mb * PB;
    const int first = (wave + (int)(wg_linear & (NW - 1))) & (NW - 1);
    // split of the last, partial round
    const int full = ntask / NW, rem = ntask - full * NW;
    int per = 1, gsz = 1;                       // waves per left-over tile, K slices actually used
    bool split = false;
    if (MAXT > 1 && rem > 0 && (NW % rem) == 0) {
        per = NW / rem;
        gsz = per < 4 ? per : 4;
        split = gsz > 1 && (L.k8 % (4 * gsz)) == 0;
    }
    f32x16 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        int task = first + i * NW, k8_lo = 0, k8_n = L.k8;
        bool active = task < ntask;
        if (split && i == full) {               // this wave's share of the left-over tiles
            const int slice = first % per;
            task = full * NW + first / per;
            active = slice < gsz;
            k8_n = L.k8 / gsz;
            k8_lo = slice * k8_n;
        }
        if (active) {
            const int mb = task / PB, pb = task - mb * PB;
            unsigned m = 0xFFFFu;
            if (L.mask) m = L.mask[(wg_linear * ntask + task) * 64 + lane];
            f32x16 c;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.0f;
            c = tile_mac<BLK, false>(L.w + ((size_t)mb * L.k8 + k8_lo) * 64 + lane, k8_n,
                                     buf + (size_t)k8_lo * BLK + (pb * 32 + j) * 8 + 4 * h, c);
            apply_bits(c, m);                                                    // mask . (a + b) = mask . a + mask . b
            acc[i] = c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        int task = first + i * NW;
        bool active = task < ntask;
        if (split && i == full) {
            task = full * NW + first / per;
            active = (first % per) == 0;        // slice 0 stores, the other slices add below
        }
        if (active) {
            const int mb = task / PB, pb = task - mb * PB;
            store_tile<P>(buf, mb, pb * 32 + j, h, acc[i]);
        }
    }
    if (split) {
        for (int sl = 1; sl < gsz; ++sl) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MAXT; ++i) {
                if (i == full && (first % per) == sl) {
                    const int task = full * NW + first / per;
                    const int mb = task / PB, pb = task - mb * PB;
                    add_tile<P>(buf, mb, pb * 32 + j, h, acc[i]);
                }
            }
        }
    }
}

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (dispatch id % 8), each with a private
// 4 MiB L2; taking logical tile (id % 8) * T/8 + id / 8 gives every XCD a contiguous run of tiles, i.e. whole
// rooms, so the rows a room's tiles gather (grouped / interpolated features) stay in ONE L2 instead of being
// fetched into all eight (measured with FETCH_SIZE: fp1 forward 124 MB -> see DESIGN.md).
__device__ __forceinline__ void xcd_tile(int &x, int &b)
{
    const unsigned gx = gridDim.x, T = gx * gridDim.y, id = blockIdx.y * gx + blockIdx.x;
    unsigned L = id;
    if ((T & 7u) == 0) L = (id & 7u) * (T >> 3) + (id >> 3);
    b = (int)(L / gx);
    x = (int)(L - (unsigned)b * gx);
}

}  // namespace psg
