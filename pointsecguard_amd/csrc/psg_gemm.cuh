// Generic row-major fp32 GEMM on the gfx950 matrix cores with fused epilogues:
//     out[r][m] = epi( sum_k in[r][k] * w[m][k] )          r < rows, m < M, k < K
// (1x1 convolution over point-major rows).  Used by the ResGCN path (EdgeConv vertex products, fusion and
// prediction layers and their input-gradient transposes, and the feature-space kNN distance matrix).
//
// Tiling: a workgroup of 4 waves owns BR rows x BN output channels; every wave owns a 64x64 sub-tile as
// 2x2 MFMA tiles (v_mfma_f32_32x32x2_f32, 4 accumulators = 64 AGPRs), so one ds_read_b128 per operand block
// feeds 8 MFMAs.  K advances in steps of 32 through LDS in the k8-block layout [k/8][row][8] (see
// psg_mlp.cuh), both operands staged by all 256 threads with 16-byte global loads.
// D orientation: D[m = channel][n = row]: lane = row, registers = channels, so the epilogue stores one
// float4 (4 consecutive channels of one row) per lane per 4 accumulator registers.
//
// ASC_K = true stores each 8-chunk permuted so that MFMA step t consumes k = 8*k8 + 2*t + h: the
// accumulation is then the ascending-k fmaf chain starting from 0 that a CPU sgemm produces
// (SURVEY.md section 8a': needed for bit-exact feature-space kNN graphs, ResGCN/gcn_lib/dense/torch_edge.py:41).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psg_mlp.cuh"

namespace psg {

enum GemmEpi { EPI_LINEAR = 0, EPI_RELU_AFFINE = 1, EPI_KNN_DIST = 2, EPI_LRELU = 3 };   // LRELU: leaky_relu(0.2), sign bits to mask_out

struct GemmArgs {
    const float *in;      // [rows][ld_in]
    const float *w;       // [M][ld_w]
    const float *bias;    // [M] or null
    const float *gbias;   // per-group bias [rows / group_rows][M] or null (broadcast concat term)
    const float *scale;   // [M] affine applied AFTER the ReLU (Conv -> ReLU -> BatchNorm order), or null
    const float *shift;   // [M]
    const float *sq;      // EPI_KNN_DIST: [rows] squared norms (same array indexes rows and columns)
    float *out;           // [rows][ld_out]
    uint32_t *mask_out;   // ReLU bits out: [rows][ceil(M/32)] words, or null
    const uint32_t *mask_in;  // multiply the result by these bits (backward through the producer's ReLU), or null
    int rows, K, M, ld_in, ld_w, ld_out, group_rows;
    int accumulate;       // 1: out += result;  2: out = addend + result (out is not read)
    const float *addend;  // with accumulate: out = (out + addend[row][c]) + result  (rows of ld_add floats), or null
    int ld_add;
    // derivative of a leaky ReLU applied to what is STORED (after accumulate / addend): out *= (bit ? 1 : post_slope).  The
    // last contributor of a gradient buffer applies the activation's derivative itself instead of a separate pass.
    const uint32_t *post_mask = nullptr;   // [rows][ceil(M/32)] words, or null
    float post_slope = 0.0f;
    // a second operand of the pre-activation sum: z = (acc + bias) + pre_add[row][c], THEN the activation (RandLA-Net's
    // residual block: leaky_relu(mlp2(x) + shortcut(y)) without a separate add pass); rows of ld_pre floats, or null
    const float *pre_add = nullptr;
    int ld_pre = 0;
};

// TQ x TI = 32-row x 32-channel MFMA tiles per wave (2 x 2 by default; 1 x 1 gives 64 x 64 workgroup tiles, i.e. four
// times as many workgroups, for the per-vertex GEMMs of a single 4096-point room that would otherwise occupy 16-32 CUs)
// (240 registers with the 64 accumulators: two waves per SIMD.  Asking the compiler for three - 168 registers, 20 bytes spilled
// outside the loop - left the fusion layer at 555 us, four spilled into the loop: 819 us; tools/gemm_probe.hip)
template <int WM, int WN, int EPI, bool ASC_K, int TQ = 2, int TI = 2>
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmArgs a)
{
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BR = 32 * TQ * WM, BN = 32 * TI * WN;
    constexpr int BLK_R = BR * 8 + 8, BLK_N = BN * 8 + 8;  // floats per 8-k block (+8: bank de-phasing)
    __shared__ float s_in[4 * BLK_R];
    __shared__ float s_w[4 * BLK_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wr = wave / WN, wc = wave % WN;
    // Tile order (round 5).  Workgroups are dispatched round-robin to the 8 XCDs, each with a private 4 MiB L2.  With the row
    // tile on blockIdx.x and the column tile on blockIdx.y an XCD met the same input rows once per column tile, a full pass of
    // the grid apart: 14.7 MB of input per XCD lay between two uses in the fusion layer (16 384 x 1792 x 1024), and the input
    // came from the fabric 2.7 x (396 MB counted per launch against 191 MB algorithmic, profiles/r04_pmc_traffic_gcn.txt).  Now
    // XCD x takes the contiguous row tiles [x R / 8, (x + 1) R / 8) and walks the COLUMN tiles of a row tile back to back: the
    // input tile is fetched once and served to its 8 column tiles from that L2; the weights (7.3 MB) are what every XCD
    // re-reads: 396 -> 317 MB per launch counted.  Same arithmetic; time within 1 % (397 against 401 us on one box: the layer
    // is matrix-bound at 0.7 TB/s).
    int row0, col0;
    {
        const unsigned nx = gridDim.x, ny = gridDim.y, T = nx * ny, id = blockIdx.y * nx + blockIdx.x;
        unsigned L = id;
        if ((T & 7u) == 0u) L = (id & 7u) * (T >> 3) + (id >> 3);
        row0 = (int)(L / ny) * BR;
        col0 = (int)(L % ny) * BN;
#ifdef PSG_GEMM_LEGACY_ORDER      // (A/B builds: tools/build_variant.sh legacy psg_resgcn "-DPSG_GEMM_LEGACY_ORDER")
        row0 = blockIdx.x * BR; col0 = blockIdx.y * BN;
#endif
    }

    f32x16 acc[TI][TQ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int q = 0; q < TQ; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.0f;

    constexpr int NI = BR * 8 / 256, NWL = BN * 8 / 256;
    // plain (non-ASC) path: the global loads of step k0 + 32 are issued right after the barrier that publishes step k0 in
    // LDS, so they are in flight under that step's MFMAs (register double buffering; one LDS buffer)
    float4 vi[NI], vw[NWL];
    // (workgroup-uniform) the whole tile lies inside both matrices and every row is 16-byte aligned: no per-piece bounds
    // tests in the k-loop - on this chip every vector instruction of the loop is paid in matrix time (fp32 MFMAs and
    // vector work of the same SIMD do not overlap), and the tests were a tenth of the loop's issue slots
    const bool interior = row0 + BR <= a.rows && col0 + BN <= a.M && (a.K & 31) == 0 && (a.ld_in & 3) == 0 && (a.ld_w & 3) == 0;
    const float *in_t = a.in + (size_t)(row0 + (tid >> 3)) * a.ld_in + 4 * (tid & 7);
    const float *w_t = a.w + (size_t)(col0 + (tid >> 3)) * a.ld_w + 4 * (tid & 7);
    auto load_plain = [&](int k0) {
        if (interior) {
#pragma unroll
            for (int u = 0; u < NI; ++u) vi[u] = *(const float4 *)(in_t + (size_t)(32 * u) * a.ld_in + k0);
#pragma unroll
            for (int u = 0; u < NWL; ++u) vw[u] = *(const float4 *)(w_t + (size_t)(32 * u) * a.ld_w + k0);
            return;
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int t = tid + u * 256, r = t >> 3, q = t & 7, k = k0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < a.rows) {
                const float *p = a.in + (size_t)(row0 + r) * a.ld_in + k;
                if (k + 3 < a.K && ((a.ld_in & 3) == 0)) v = *(const float4 *)p;
                else {
                    if (k < a.K) v.x = p[0];
                    if (k + 1 < a.K) v.y = p[1];
                    if (k + 2 < a.K) v.z = p[2];
                    if (k + 3 < a.K) v.w = p[3];
                }
            }
            vi[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NWL; ++u) {
            const int t = tid + u * 256, r = t >> 3, q = t & 7, k = k0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col0 + r < a.M) {
                const float *p = a.w + (size_t)(col0 + r) * a.ld_w + k;
                if (k + 3 < a.K && ((a.ld_w & 3) == 0)) v = *(const float4 *)p;
                else {
                    if (k < a.K) v.x = p[0];
                    if (k + 1 < a.K) v.y = p[1];
                    if (k + 2 < a.K) v.z = p[2];
                    if (k + 3 < a.K) v.w = p[3];
                }
            }
            vw[u] = v;
        }
    };
    if constexpr (!ASC_K) load_plain(0);

    for (int k0 = 0; k0 < a.K; k0 += 32) {
        // ---- stage in[row0.., k0..k0+31] and w[col0.., k0..k0+31] (zero filled outside the matrices).  All global
        // loads of the step are issued into registers first, then written to LDS: otherwise every 16-byte piece
        // pays its own round trip (a 16-workgroup GEMM went from 34 us to a few us with this).
        if constexpr (ASC_K) {
            // ascending-k layout: one thread moves a whole 8-chunk (two 16-byte loads) and writes it as two
            // ds_write_b128 {k0,k2,k4,k6} {k1,k3,k5,k7} (the first version scattered it with eight ds_write_b32)
            constexpr int CI = BR * 4 / 256, CW = BN * 4 / 256;
            float4 ia[CI][2], wa2[CW][2];
#pragma unroll
            for (int u = 0; u < CI; ++u) {
                const int t = tid + u * 256, r = t >> 2, c8 = t & 3, k = k0 + 8 * c8;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (row0 + r < a.rows) {
                    const float *p = a.in + (size_t)(row0 + r) * a.ld_in + k;
                    if (k + 7 < a.K && ((a.ld_in & 3) == 0)) { v0 = *(const float4 *)p; v1 = *(const float4 *)(p + 4); }
                    else {
                        float e[8];
#pragma unroll
                        for (int z = 0; z < 8; ++z) e[z] = k + z < a.K ? p[z] : 0.f;
                        v0 = make_float4(e[0], e[1], e[2], e[3]); v1 = make_float4(e[4], e[5], e[6], e[7]);
                    }
                }
                ia[u][0] = v0; ia[u][1] = v1;
            }
#pragma unroll
            for (int u = 0; u < CW; ++u) {
                const int t = tid + u * 256, r = t >> 2, c8 = t & 3, k = k0 + 8 * c8;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (col0 + r < a.M) {
                    const float *p = a.w + (size_t)(col0 + r) * a.ld_w + k;
                    if (k + 7 < a.K && ((a.ld_w & 3) == 0)) { v0 = *(const float4 *)p; v1 = *(const float4 *)(p + 4); }
                    else {
                        float e[8];
#pragma unroll
                        for (int z = 0; z < 8; ++z) e[z] = k + z < a.K ? p[z] : 0.f;
                        v0 = make_float4(e[0], e[1], e[2], e[3]); v1 = make_float4(e[4], e[5], e[6], e[7]);
                    }
                }
                wa2[u][0] = v0; wa2[u][1] = v1;
            }
#pragma unroll
            for (int u = 0; u < CI; ++u) {
                const int t = tid + u * 256, r = t >> 2, c8 = t & 3;
                float *d = s_in + c8 * BLK_R + r * 8;     // element k%8 = 2t+h goes to slot 4h+t
                *(float4 *)d = make_float4(ia[u][0].x, ia[u][0].z, ia[u][1].x, ia[u][1].z);
                *(float4 *)(d + 4) = make_float4(ia[u][0].y, ia[u][0].w, ia[u][1].y, ia[u][1].w);
            }
#pragma unroll
            for (int u = 0; u < CW; ++u) {
                const int t = tid + u * 256, r = t >> 2, c8 = t & 3;
                float *d = s_w + c8 * BLK_N + r * 8;
                *(float4 *)d = make_float4(wa2[u][0].x, wa2[u][0].z, wa2[u][1].x, wa2[u][1].z);
                *(float4 *)(d + 4) = make_float4(wa2[u][0].y, wa2[u][0].w, wa2[u][1].y, wa2[u][1].w);
            }
        } else {
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int t = tid + u * 256, r = t >> 3, q = t & 7;
            const float4 v = vi[u];
            float *d = s_in + (q >> 1) * BLK_R + r * 8;
            if (ASC_K) {  // element k%8 = 2t+h goes to slot 4h+t
                const int e = (q & 1) * 4;
                d[((e + 0) & 1) * 4 + ((e + 0) >> 1)] = v.x;
                d[((e + 1) & 1) * 4 + ((e + 1) >> 1)] = v.y;
                d[((e + 2) & 1) * 4 + ((e + 2) >> 1)] = v.z;
                d[((e + 3) & 1) * 4 + ((e + 3) >> 1)] = v.w;
            } else {
                *(float4 *)(d + (q & 1) * 4) = v;
            }
        }
#pragma unroll
        for (int u = 0; u < NWL; ++u) {
            const int t = tid + u * 256, r = t >> 3, q = t & 7;
            const float4 v = vw[u];
            float *d = s_w + (q >> 1) * BLK_N + r * 8;
            if (ASC_K) {
                const int e = (q & 1) * 4;
                d[((e + 0) & 1) * 4 + ((e + 0) >> 1)] = v.x;
                d[((e + 1) & 1) * 4 + ((e + 1) >> 1)] = v.y;
                d[((e + 2) & 1) * 4 + ((e + 2) >> 1)] = v.z;
                d[((e + 3) & 1) * 4 + ((e + 3) >> 1)] = v.w;
            } else {
                *(float4 *)(d + (q & 1) * 4) = v;
            }
        }
        }
        __syncthreads();
        if constexpr (!ASC_K) {
            if (k0 + 32 < a.K) load_plain(k0 + 32);
        }
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            float4 wa[TI], xb[TQ];
#pragma unroll
            for (int i = 0; i < TI; ++i) wa[i] = *(const float4 *)(s_w + k8 * BLK_N + (wc * 32 * TI + i * 32 + j) * 8 + 4 * h);
#pragma unroll
            for (int q = 0; q < TQ; ++q) xb[q] = *(const float4 *)(s_in + k8 * BLK_R + (wr * 32 * TQ + q * 32 + j) * 8 + 4 * h);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < TQ; ++q) acc[i][q] = mfma4<false>(wa[i], xb[q], acc[i][q]);
        }
        __syncthreads();
    }

    // ---- epilogue: lane (j,h) of tile (i,q) holds channels cbase + 8g + 4h + (0..3) of row rbase + j
    if (EPI != EPI_KNN_DIST && interior && (a.ld_out & 3) == 0 && (!a.addend || (a.ld_add & 3) == 0) && (!a.pre_add || (a.ld_pre & 3) == 0)) {
        // interior tiles: the per-channel terms are read once per 32-channel tile as float4 (they were 3 scalar loads per
        // value: the epilogue of the 16 384 x 1792 x 1024 fusion layer cost a tenth of its launch), same operations in
        // the same order as the general path below
        const int nw = (a.M + 31) >> 5;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int cbase = col0 + wc * 32 * TI + i * 32;
            float4 b4[4], s4[4], t4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cbase + 8 * g + 4 * h;
                b4[g] = a.bias ? *(const float4 *)(a.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                s4[g] = a.scale ? *(const float4 *)(a.scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
                t4[g] = a.scale ? *(const float4 *)(a.shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < TQ; ++q) {
                const int row = row0 + wr * 32 * TQ + q * 32 + j;
                const unsigned min_bits = a.mask_in ? a.mask_in[(size_t)row * nw + (cbase >> 5)] : 0xFFFFFFFFu;
                const unsigned post_bits = a.post_mask ? a.post_mask[(size_t)row * nw + (cbase >> 5)] : 0xFFFFFFFFu;
                const float *gb = a.gbias ? a.gbias + (size_t)(row / a.group_rows) * a.M + cbase + 4 * h : nullptr;
                const float *pa = (a.pre_add && (a.ld_pre & 3) == 0) ? a.pre_add + (size_t)row * a.ld_pre + cbase + 4 * h : nullptr;
                float *o = a.out + (size_t)row * a.ld_out + cbase + 4 * h;
                unsigned mbits = 0;
                float4 vv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 g4 = gb ? *(const float4 *)(gb + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 p4 = pa ? *(const float4 *)(pa + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float bb[4] = {b4[g].x, b4[g].y, b4[g].z, b4[g].w}, ss[4] = {s4[g].x, s4[g].y, s4[g].z, s4[g].w};
                    const float tt[4] = {t4[g].x, t4[g].y, t4[g].z, t4[g].w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
                    float e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int cl = 8 * g + 4 * h + u;
                        float z = acc[i][q][4 * g + u];
                        if (a.bias) z += bb[u];
                        if (gb) z += gg[u];
                        if (pa) z += u == 0 ? p4.x : (u == 1 ? p4.y : (u == 2 ? p4.z : p4.w));
                        if (EPI == EPI_RELU_AFFINE) {
                            const bool pos = z > 0.0f;
                            mbits |= (unsigned)pos << cl;
                            z = pos ? z : 0.0f;
                            if (a.scale) z = z * ss[u] + tt[u];
                        }
                        if (EPI == EPI_LRELU) {
                            const bool pos = z > 0.0f;
                            mbits |= (unsigned)pos << cl;
                            z = pos ? z : 0.2f * z;
                        }
                        if (!((min_bits >> cl) & 1u)) z = 0.0f;
                        e[u] = z;
                    }
                    vv[g] = make_float4(e[0], e[1], e[2], e[3]);
                }
                if ((EPI == EPI_RELU_AFFINE || EPI == EPI_LRELU) && a.mask_out) {
                    const unsigned other = __shfl_xor(mbits, 32);
                    if (h == 0) a.mask_out[(size_t)row * nw + (cbase >> 5)] = mbits | other;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = vv[g];
                    float4 *dst = (float4 *)(o + 8 * g);
                    if (a.accumulate) {
                        float4 old = a.accumulate == 1 ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
                        if (a.addend) {
                            const float4 ad = *(const float4 *)(a.addend + (size_t)row * a.ld_add + cbase + 8 * g + 4 * h);
                            old.x += ad.x; old.y += ad.y; old.z += ad.z; old.w += ad.w;
                        }
                        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                    }
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
        return;
    }
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        const int row = row0 + wr * 32 * TQ + q * 32 + j;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int cbase = col0 + wc * 32 * TI + i * 32;
            unsigned mbits = 0;
            unsigned min_bits = 0xFFFFFFFFu;
            if (a.mask_in && row < a.rows && cbase < a.M) min_bits = a.mask_in[(size_t)row * ((a.M + 31) >> 5) + (cbase >> 5)];
            unsigned post_bits = 0xFFFFFFFFu;
            if (a.post_mask && row < a.rows && cbase < a.M) post_bits = a.post_mask[(size_t)row * ((a.M + 31) >> 5) + (cbase >> 5)];
            float vals[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = acc_row(r, h);  // channel within the 32-tile
                const int c = cbase + cl;
                float z = acc[i][q][r];
                if (EPI == EPI_KNN_DIST) {
                    // (|x_i|^2 + (-2 x_i.x_j)) + |x_j|^2, torch_edge.py:41-43 (i = row, j = column)
                    const float si = row < a.rows ? a.sq[row] : 0.f, sj = c < a.M ? a.sq[c] : 0.f;
                    z = __fadd_rn(__fadd_rn(si, __fmul_rn(-2.0f, z)), sj);
                } else {
                    if (a.bias && c < a.M) z += a.bias[c];
                    if (a.gbias && c < a.M && row < a.rows) z += a.gbias[(size_t)(row / a.group_rows) * a.M + c];
                    if (a.pre_add && c < a.M && row < a.rows) z += a.pre_add[(size_t)row * a.ld_pre + c];
                    if (EPI == EPI_RELU_AFFINE) {
                        const bool pos = z > 0.0f;
                        mbits |= (unsigned)pos << cl;
                        z = pos ? z : 0.0f;
                        if (a.scale && c < a.M) z = z * a.scale[c] + a.shift[c];
                    }
                    if (EPI == EPI_LRELU) {
                        const bool pos = z > 0.0f;
                        mbits |= (unsigned)pos << cl;
                        z = pos ? z : 0.2f * z;
                    }
                    if (!((min_bits >> cl) & 1u)) z = 0.0f;
                }
                vals[r] = z;
            }
            if ((EPI == EPI_RELU_AFFINE || EPI == EPI_LRELU) && a.mask_out) {
                const unsigned other = __shfl_xor(mbits, 32);
                if (h == 0 && row < a.rows && cbase < a.M) a.mask_out[(size_t)row * ((a.M + 31) >> 5) + (cbase >> 5)] = mbits | other;
            }
            if (row < a.rows) {
                float *o = a.out + (size_t)row * a.ld_out + cbase + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = cbase + 8 * g + 4 * h;
                    if (c + 3 < a.M && ((a.ld_out & 3) == 0)) {
                        float4 v = make_float4(vals[4 * g], vals[4 * g + 1], vals[4 * g + 2], vals[4 * g + 3]);
                        float4 *dst = (float4 *)(o + 8 * g);
                        if (a.accumulate) {
                            float4 old = a.accumulate == 1 ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
                            if (a.addend) {
                                const float4 ad = *(const float4 *)(a.addend + (size_t)row * a.ld_add + c);
                                old.x += ad.x; old.y += ad.y; old.z += ad.z; old.w += ad.w;
                            }
                            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                        }
                        if (a.post_mask) {
                            const unsigned pb = post_bits >> (8 * g + 4 * h);
                            if (!(pb & 1u)) v.x *= a.post_slope;
                            if (!(pb & 2u)) v.y *= a.post_slope;
                            if (!(pb & 4u)) v.z *= a.post_slope;
                            if (!(pb & 8u)) v.w *= a.post_slope;
                        }
                        *dst = v;
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (c + u < a.M) {
                                float *dst = o + 8 * g + u;
                                float base = a.accumulate == 1 ? *dst : 0.0f;
                                if (a.accumulate && a.addend) base += a.addend[(size_t)row * a.ld_add + c + u];
                                float vv = a.accumulate ? vals[4 * g + u] + base : vals[4 * g + u];
                                if (a.post_mask && !((post_bits >> (8 * g + 4 * h + u)) & 1u)) vv *= a.post_slope;
                                *dst = vv;
                            }
                    }
                }
            }
        }
    }
}

}  // namespace psg
