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

// Timing-diagnosis switches (tools/diag_*.sh: skip a kernel section to see what it costs; the results are then wrong).
// They exist ONLY in a diagnostic build (make EXTRA=-DPSG_DIAG_BUILD): in the default library every test below is the
// constant 0, the compiler removes the branches, and no environment variable can make a kernel skip its work.
#ifdef PSG_DIAG_BUILD
#define PSG_DIAGBIT(a, bit) ((a).diag & (bit))
#else
#define PSG_DIAGBIT(a, bit) 0
#endif

namespace psg {

struct SaFwdArgs {
    const float *xyz;      // [B][Np][xyz_stride], first 3 floats of a row = xyz
    const float *feat;     // [B][Np][D]
    const float *new_xyz;  // [B][S][3]
    const int32_t *gidx;   // [B][S][KS]
    float *out;            // [B][S][ld_out], this module's C3 channels at column c_out (MSG scales share a row)
    uint8_t *arg;          // [B][S][C3] arg-max sample, 255 = no gradient (max <= 0)
    FwdLayer l1, l2;
    const float4 *w3;      // last layer, packed like FwdLayer::w, used as the B operand (flipped tile)
    const float *b3;
    int k8_3, nb3;
    int xyz_stride, D, Np, S, C3;
    int ld_out, c_out;
    // SPLIT kernels (see sa_fwd_kernel): per-point first-layer products T = feats . W1f^T + b1, rows [B][Np][ldt]; l1 then
    // holds the xyz columns of the first layer only (k8 = 1)
    const float *tfeat;
    int ldt;
    int diag;              // timing diagnostics only (-DPSG_DIAG_BUILD libraries only): skip sections, results are then wrong
};

struct SaBwdArgs {
    // The pooled-output gradient of this module is columns [c_off, c_off + C3) of rows of `ld` floats (ld = all
    // channels of the level: an MSG level concatenates its scales); dout, dint and gsa rows all have that stride.
    const float *dout;    // [B][S][ld] (direct), or null when the gradient is gathered through nninv_*
    // gather form: dout[s][c] = sum over the fine points p whose 3-NN lists contain s of w * dint[p][c]
    const int32_t *nninv_off;   // [B][S+1]
    const int2 *nninv_ent;      // [B][3*n_fine] {fine point, weight bits}, sorted by fine point per list
    const float *dint;          // [B][n_fine][C3]
    int n_fine;
    // grouping-transpose form: dout[s][c] += sum over the grouped rows e of the next SA level that gathered point s
    const int32_t *ginv_off;    // [B][S+1] or null: list of point s = rows [off[s], off[s+1]) of gsa
    const float *gsa;           // [B][g_rows][ld] gradient rows written by the next level's sa_bwd IN LIST ORDER
    int g_rows;
    const int32_t *ginv_off2;   // second scale of the next level (MSG), or null
    const float *gsa2;
    int g_rows2;
    int ld, c_off;
    float *gsa_out;             // [B][S*KS][cg_out]: this module's grouped-input gradient rows (plain stores), row
    const int32_t *gpos_out;    // (group*KS + sample) stored at position gpos_out[row] = its slot in the consumer's lists
    int cg_out;
    const uint8_t *arg;   // [B][S][C3]
    const int32_t *gidx;  // [B][S][KS]
    BwdLayer l3t, l2t, l1t;
    int D, Np, S, C3;
    int c_lo, c_hi;       // feature channels [c_lo, c_hi) of the grouped-input gradient are scattered
    int split;            // 1: the rows stored are dZ1 (the first layer's pre-activation gradient, channels [c_lo, c_hi) =
                          // [0, C1)), the first layer's transpose is NOT applied here (pw_bwd_kernel applies it per POINT)
    int dsrc_blk;         // LDS block where the gathered pooled-output gradient is staged
    // colour-only request at level 0 (the attack loop): the first layer's feature columns, plain rows [C1][D] (w1c; null: the
    // transposed layer runs on the matrix pipe as everywhere else); the three wanted columns are staged behind dsrc
    const float *w1c;
    int C1, w1c_off;      // first layer's width (16 or 32); float offset of the LDS copy from the start of the staging block
    int pos_off;          // float offset (staging block) of the LDS copy of the workgroup's P output slots
    int diag;             // timing diagnostics only (-DPSG_DIAG_BUILD libraries only)
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
    // FP split (see fp_layer1_split).  Consumer side: tsrc = rows [B][S][ldt] of the interpolated part's first-layer product,
    // computed per COARSE point by the coarser module; layer[0] then holds the skip columns only (k8 = C1 / 8, possibly 0).
    const float *tsrc;
    int ldt;
    // Producer side: after its own last layer the module runs `extra` (the finer module's interpolated-part columns, no bias,
    // no ReLU) over its output and writes those rows to out2 [B][N][Cout2]
    float *out2;
    int Cout2;
    FwdLayer extra;
    int diag;               // timing diagnostics only (-DPSG_DIAG_BUILD libraries only): skip sections, results are then wrong
    unsigned long long *dbg; // diag & 256: per-workgroup {memtime, memrealtime} at entry and exit
};

struct FpBwdArgs {
    const float *dout;          // [B][N][Cout] gradient of the last layer's (post-ReLU) output, or null
    // gather form of dout (deterministic transpose of the finer module's 3-NN interpolation):
    const int32_t *nninv_off;   // [B][N+1] or null
    const int2 *nninv_ent;      // [B][3*n_fine]
    const float *dint;          // [B][n_fine][Cg] interpolated-part gradient rows written by the finer module
    int n_fine;
    int Cg;                     // = Cout, or (finer module split) the width of ITS first layer: the rows are its dZ1
    BwdLayer pre;               // finer module split (pre.w != null): gathered rows -> . W1b(finer)^T, masked by mask_last
    float *dint_out;            // [B][N][C2]: this module's interpolated-part gradient rows (plain stores); split: its dZ1
                                // rows [B][N][Cd], written BEFORE the skip columns' transpose `skipT` (C1 > 0) runs
    int split, Cd;
    BwdLayer skipT;
    const uint16_t *mask_last;  // ReLU mask of that layer
    const float *logp;          // head mode: [B][N][n_cls]
    const float *dlogp;         // head mode: [B][N][n_cls]
    const int32_t *nn_idx;
    const float *nn_w;
    float *dfeat1;              // [B][N][C1] skip-link gradient rows (plain stores) or null
    BwdLayer layer[MAX_LAYERS];
    int n_layers;
    int C1, C2, N, S, Cout, n_cls, mb_last;
};

// ------------------------------------------------------------------------------------------ SA fwd
// LDS channel order of the grouped input: [feats(D), rel_xyz(3), zero pad to a multiple of 8]; the
// first layer's weight columns are permuted accordingly at pack time (reference order is
// [rel_xyz, feats], pointnet_util.py:137).
// KS = samples per group (32, or 16 for the small-radius scale of an MSG level: two groups per 32-point tile);
// MAXT = tiles a wave may hold in the first two layers.
//
// SPLIT (round 5): the first layer is linear in the concatenation, W1 . [x_j - c_i ; f_j] = W1f . f_j + W1x . (x_j - c_i),
// and its feature part depends on the SOURCE POINT j only - a level has 8 x fewer points than grouped rows (1024 / 256 / 64
// points against 8192 / 2048 / 512 rows per room at levels 1 - 3).  T[j] = W1f . f_j + b1 is computed once per point by a
// separate per-point launch; here the rows of T are gathered through the group table straight into the MFMA ACCUMULATORS
// of the first layer's tiles (lane (j, h) of a tile reads the 4 x 16 bytes of row gidx[j] that its accumulator registers
// hold: the same bytes per grouped row as the feature gather it replaces, and no LDS staging), and the layer itself is
// ONE k8-chunk (the three relative coordinates) instead of (D + 3) / 8 chunks: 9 -> 1, 17 -> 1, 33 -> 1 at levels 1 - 3,
// i.e. 24 % of the module's matrix instructions.  The coordinates are NOT split (W1x x_j - W1x c_i would cancel: |x| ~ 3
// against |x_j - c_i| < r): the difference is formed first, as the reference does.
template <int P, int NW, int MAXT, int KS>
__device__ __forceinline__ void sa_layer1_split(const SaFwdArgs &a, int b, int s0, float *__restrict__ buf, size_t wg_linear)
{
    constexpr int PB = P / 32;
    const FwdLayer &L = a.l1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntask = L.mb * PB;
    const int first = (wave + (int)(wg_linear & (NW - 1))) & (NW - 1);
    const int32_t *gi = a.gidx + ((size_t)b * a.S + s0) * KS;          // the workgroup's P grouped rows are contiguous
    const float *trows = a.tfeat + (size_t)b * a.Np * a.ldt;
    f32x16 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int task = first + i * NW;
        if (task < ntask) {
            const int mb = task / PB, pb = task - mb * PB;
            const int src = gi[pb * 32 + j];
            // accumulator registers 4g .. 4g+3 of lane (j, h) are channels mb*32 + 8g + 4h + (0..3) of point j
            const float4 *tp = (const float4 *)(trows + __umul24((unsigned)src, (unsigned)a.ldt) + mb * 32 + 4 * h);
            const float4 t0 = tp[0], t1 = tp[2], t2 = tp[4], t3 = tp[6];
            f32x16 c;
            c[0] = t0.x; c[1] = t0.y; c[2] = t0.z; c[3] = t0.w;
            c[4] = t1.x; c[5] = t1.y; c[6] = t1.z; c[7] = t1.w;
            c[8] = t2.x; c[9] = t2.y; c[10] = t2.z; c[11] = t2.w;
            c[12] = t3.x; c[13] = t3.y; c[14] = t3.z; c[15] = t3.w;
            const float4 wx = L.w[(size_t)mb * 64 + lane];               // k8 = 1: one chunk per 32-channel block
            const float4 rx = *(const float4 *)(buf + (pb * 32 + j) * 8 + 4 * h);
            c = mfma4<false>(wx, rx, c);
            mfma_fence(c);                                               // (compiler MFMAs -> inline-asm consumer: psg_mlp.cuh)
            const unsigned m = relu_bits(c);
            if (L.mask) L.mask[(wg_linear * ntask + task) * 64 + lane] = (uint16_t)m;
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

template <int P, int NW, int KS = 32, int MAXT = 1, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64) void sa_fwd_kernel(SaFwdArgs a)
{
    using L = Lds<P>;
    static_assert(KS == 32 || KS == 16, "groups of 32 or 16 samples");
    constexpr int G = P / KS, PB = P / 32, NT = NW * 64, NPART = NT / P;
    extern __shared__ float lds[];
    float *buf0 = lds;   // the one activation buffer (layers run in place)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, b;
    xcd_tile(bx, b);
    const int s0 = bx * G;
    const size_t wg = (size_t)b * gridDim.x + bx;

    if (SPLIT) {
        // block 0 of the buffer = [x_j - c_i, 0 x 5] of the P grouped rows; the feature part arrives through T (below)
        if (tid < P) {
            const int j = tid, s = s0 + j / KS;
            const int src = a.gidx[((size_t)b * a.S + s) * KS + (j & (KS - 1))];
            const float *xr = a.xyz + ((size_t)b * a.Np + src) * a.xyz_stride;
            const float *cr = a.new_xyz + ((size_t)b * a.S + s) * 3;
            *(float4 *)(buf0 + j * 8) = make_float4(xr[0] - cr[0], xr[1] - cr[1], xr[2] - cr[2], 0.0f);
            *(float4 *)(buf0 + j * 8 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else if (!PSG_DIAGBIT(a, 1)) {   // gather the P grouped points
        const int j = tid % P, part = tid / P;
        const int s = s0 + j / KS;
        const int src = a.gidx[((size_t)b * a.S + s) * KS + (j & (KS - 1))];
        const float *frow = a.feat + ((size_t)b * a.Np + src) * a.D;
        const float *xr = a.xyz + ((size_t)b * a.Np + src) * a.xyz_stride;
        const float *cr = a.new_xyz + ((size_t)b * a.S + s) * 3;
        if ((a.D & 3) == 0) {
            const float4 *f4 = (const float4 *)frow;
            for (int q = part; q < (a.D >> 2); q += NPART) *(float4 *)(buf0 + L::off(4 * q, j)) = f4[q];
            if (part == 0) {
                *(float4 *)(buf0 + L::off(a.D, j)) = make_float4(xr[0] - cr[0], xr[1] - cr[1], xr[2] - cr[2], 0.0f);
                if ((a.D & 7) == 0) *(float4 *)(buf0 + L::off(a.D + 4, j)) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (part == 0) {
            // sa1: D = 9 -> [f0..f8, rx, ry, rz, 0, 0, 0, 0] (two 8-channel blocks)
            float f[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) f[c] = frow[c];
            *(float4 *)(buf0 + L::off(0, j)) = make_float4(f[0], f[1], f[2], f[3]);
            *(float4 *)(buf0 + L::off(4, j)) = make_float4(f[4], f[5], f[6], f[7]);
            *(float4 *)(buf0 + L::off(8, j)) = make_float4(f[8], xr[0] - cr[0], xr[1] - cr[1], xr[2] - cr[2]);
            *(float4 *)(buf0 + L::off(12, j)) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // zero any further 8-channel blocks the first layer reads (its K is ceil((D + 3) / 8) blocks: normally none)
        for (int blk = ((a.D + 3) >> 3) + 1 + part; blk < a.l1.k8; blk += NPART) {
            float *z = buf0 + (size_t)blk * L::BLK + j * 8;
            *(float4 *)z = make_float4(0.f, 0.f, 0.f, 0.f);
            *(float4 *)(z + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    // biases of this wave's last-layer tiles: fetched now, so their latency is spent under the first two layers
    // (an SSG level has nb3 * PB = 2 * NW last-layer tiles, i.e. two per wave; MSG scales have at most that)
    const int jj = lane & 31, h = lane >> 5;
    constexpr int T3 = 2;
    float bias3[T3];
#pragma unroll
    for (int i = 0; i < T3; ++i) {
        const int task = wave + i * NW;
        bias3[i] = task < a.nb3 * PB ? a.b3[(task / PB) * 32 + jj] : 0.0f;
    }
    if (SPLIT) sa_layer1_split<P, NW, MAXT, KS>(a, b, s0, buf0, wg);
    else if (!PSG_DIAGBIT(a, 8)) layer_fwd<P, NW, MAXT>(a.l1, buf0, wg);
    if (!PSG_DIAGBIT(a, 16)) __syncthreads();
    if (!PSG_DIAGBIT(a, 8)) layer_fwd<P, NW, MAXT>(a.l2, buf0, wg);
    if (!PSG_DIAGBIT(a, 16)) __syncthreads();
    if (PSG_DIAGBIT(a, 32)) return;

    // last layer with the tile flipped (D[point][channel]) so the max over the 32 samples of a
    // group is an in-lane max over 16 accumulators + one exchange between lane halves.
#pragma unroll
    for (int i = 0; i < T3; ++i) {
        const int task = wave + i * NW;
        if (task >= a.nb3 * PB) break;
        const int nb = task / PB, g = task - nb * PB;   // g = 32-point tile of the workgroup
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias3[i];
        acc = tile_mac<L::BLK, true>(a.w3 + (size_t)nb * a.k8_3 * 64 + lane, a.k8_3, buf0 + (g * 32 + jj) * 8 + 4 * h,
                                     acc);
        if (KS == 32) {
            // max(relu(x)) = relu(max(x)): the raw maximum first (v_max3: 8 instructions for 16 values), then the lowest row
            // that holds it (descending scan: the last match written is the lowest), the ReLU once at the end - the values and
            // arg bytes of the per-element compare / select chain (80 instructions), in about half of them
            float best = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) best = fmaxf(fmaxf(best, acc[r]), acc[r + 1]);
            best = fmaxf(best, acc[15]);
            int bidx = acc_row(15, h);
#pragma unroll
            for (int r = 14; r >= 0; --r) bidx = acc[r] == best ? acc_row(r, h) : bidx;
            best = best > 0.0f ? best : 0.0f;                 // (+0 for a non-positive maximum, as the element-wise ReLU gave)
            float ob = __shfl_xor(best, 32);
            int oi = __shfl_xor(bidx, 32);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
            if (h == 0) {
                const size_t row = (size_t)b * a.S + s0 + g;
                a.out[row * a.ld_out + a.c_out + nb * 32 + jj] = best;
                a.arg[row * a.C3 + nb * 32 + jj] = best > 0.0f ? (uint8_t)bidx : (uint8_t)255;
            }
        } else {
            // two groups of 16 samples per tile: accumulators 0..7 hold points 0..15, 8..15 hold points 16..31
            float best[2] = {-1.0f, -1.0f};
            int bidx[2] = {0, 0};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[r] > 0.0f ? acc[r] : 0.0f;
                if (v > best[r >> 3]) { best[r >> 3] = v; bidx[r >> 3] = acc_row(r, h) & 15; }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float ob = __shfl_xor(best[q], 32);
                int oi = __shfl_xor(bidx[q], 32);
                if (ob > best[q] || (ob == best[q] && oi < bidx[q])) { best[q] = ob; bidx[q] = oi; }
            }
            // lane half h writes group h of the tile
            const float bv = h ? best[1] : best[0];
            const int bi = h ? bidx[1] : bidx[0];
            const size_t row = (size_t)b * a.S + s0 + 2 * g + h;
            a.out[row * a.ld_out + a.c_out + nb * 32 + jj] = bv;
            a.arg[row * a.C3 + nb * 32 + jj] = bv > 0.0f ? (uint8_t)bi : (uint8_t)255;
        }
    }
}

// ------------------------------------------------------------------------------------------ SA bwd
// Level 0 of the attack loop wants THREE of the first layer's twelve input-gradient channels (the colours).  As a transposed
// MFMA layer that is a 32-row output tile per 32 points, sixteen matrix instructions and a barrier for 3 useful rows, then a
// pass that picks them out of LDS again.  Here: the lanes of a wave take its 32 grouped rows twice - lane half h sums the
// channels [h C1 / 2, (h + 1) C1 / 2) of dZ1 against the three colour columns (a 3 x C1 table in LDS, read as broadcasts), the
// halves are added across the wave and lane half 0 stores the compact {r, g, b, 0} row straight from registers.
// 3 x C1 MACs per row on the vector pipe instead of 32 x C1 on the matrix pipe (sa1 backward 7.8 -> 6.9 ms per attack).
template <int P, int NW, int C1>
__device__ __forceinline__ void sa_l1t_colour(const SaBwdArgs &a, const float *__restrict__ buf0, const float *__restrict__ tab,
                                              const int32_t *__restrict__ pos, float *__restrict__ orow)
{
    using L = Lds<P>;
    static_assert(P == 32 * NW, "one 32-row tile per wave");
    constexpr int KH = C1 / 2;                           // channels per lane half: 8 or 16
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr = lane & 31, h = lane >> 5, j = 32 * wave + rr;
    const int pj = pos[j];
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    const float *t = tab + h * 3 * KH;
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
        const int c = h * KH + 4 * q;
        const float4 z = *(const float4 *)(buf0 + L::off(c, j));
        const float4 w0 = *(const float4 *)(t + 4 * q), w1 = *(const float4 *)(t + KH + 4 * q), w2 = *(const float4 *)(t + 2 * KH + 4 * q);
        g0 = fmaf(z.x, w0.x, g0); g0 = fmaf(z.y, w0.y, g0); g0 = fmaf(z.z, w0.z, g0); g0 = fmaf(z.w, w0.w, g0);
        g1 = fmaf(z.x, w1.x, g1); g1 = fmaf(z.y, w1.y, g1); g1 = fmaf(z.z, w1.z, g1); g1 = fmaf(z.w, w1.w, g1);
        g2 = fmaf(z.x, w2.x, g2); g2 = fmaf(z.y, w2.y, g2); g2 = fmaf(z.z, w2.z, g2); g2 = fmaf(z.w, w2.w, g2);
    }
    // lower channels first: (sum over half 0) + (sum over half 1), the same order in both lanes of a pair
    const float o0 = __shfl_xor(g0, 32), o1 = __shfl_xor(g1, 32), o2 = __shfl_xor(g2, 32);
    if (h == 0 && pj >= 0) *(float4 *)(orow + (size_t)pj * 4) = make_float4(g0 + o0, g1 + o1, g2 + o2, 0.0f);   // padding rows are not listed
}

template <int P, int NW, int MAXT, int KS = 32>
__global__ __launch_bounds__(NW * 64) void sa_bwd_kernel(SaBwdArgs a)
{
    using L = Lds<P>;
    static_assert(KS == 32 || KS == 16, "groups of 32 or 16 samples");
    constexpr int G = P / KS, NT = NW * 64;
    extern __shared__ float lds[];
    float *buf0 = lds;   // the one activation buffer (layers run in place)
    const int tid = threadIdx.x;
    int bx, b;
    xcd_tile(bx, b);
    const int s0 = bx * G;
    const size_t wg = (size_t)b * gridDim.x + bx;

    // gradient of the pooled output: read directly, or gathered (no atomics, fixed summation order) from the
    // interpolated-part rows of the feature-propagation module that upsampled this level
    // arg-max bytes of this workgroup's groups: issued first, so their latency overlaps the gradient gather below
    const int nblk = a.C3 >> 3;
    constexpr int NTASK = 4;   // (point, 8-channel block) tasks per thread: P * C3 / 8 / NT = 4 for every SSG level (<= 4 for MSG)
    uint2 am_pre[NTASK];
#pragma unroll
    for (int i = 0; i < NTASK; ++i) {
        const int t = tid + i * NT;
        am_pre[i] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        if (t < P * nblk && !PSG_DIAGBIT(a, 1)) {
            const int pnt = t % P, blk = t / P;
            am_pre[i] = *(const uint2 *)(a.arg + ((size_t)b * a.S + s0 + pnt / KS) * a.C3 + blk * 8);
        }
    }
    // dout[s][c] = skip-link gradient (plain rows) + transposed 3-NN interpolation + transposed grouping of the
    // next level, every sum in a fixed order (ascending fine point / grouped row)
    float *dsrc = lds + (size_t)a.dsrc_blk * L::BLK;   // staging block(s) behind the activation buffer
    // the rows' slots in the consumer's lists: read now (one coalesced load), used after the last layer - fetched there they
    // were a cache-line miss at the tail of every workgroup
    int32_t *s_pos = (int32_t *)(dsrc + a.pos_off);
    if (!a.w1c && tid < P) s_pos[tid] = a.gpos_out[(size_t)b * a.S * KS + (size_t)s0 * KS + tid];   // (the colour path reads its one slot per lane itself)
    if (a.w1c && tid < 3 * a.C1) {   // table [half][column][C1 / 2] of the wanted columns (read after the barriers of the layers below)
        const int kh = a.C1 >> 1, hh = tid / (3 * kh), r = (tid / kh) % 3, cc = tid % kh;
        dsrc[a.w1c_off + tid] = a.w1c[(hh * kh + cc) * a.D + a.c_lo + r];
    }
    for (int t = tid; t < G * a.C3; t += NT) {
        const int g = t / a.C3, c = t - g * a.C3;
        const int cc = a.c_off + c;
        float acc = a.dout ? a.dout[((size_t)b * a.S + s0 + g) * a.ld + cc] : 0.0f;
        if (a.nninv_off) {
            const int32_t *off = a.nninv_off + (size_t)b * (a.S + 1) + s0 + g;
            const int2 *ent = a.nninv_ent + (size_t)b * 3 * a.n_fine;
            const float *drows = a.dint + (size_t)b * a.n_fine * a.ld + cc;
            const int e1 = off[1];
            int e = off[0];
            for (; e + 4 <= e1; e += 4) {   // four entries and their rows in flight; added in list order
                const int2 p0 = ent[e], p1 = ent[e + 1], p2 = ent[e + 2], p3 = ent[e + 3];
                const float v0 = drows[(size_t)p0.x * a.ld], v1 = drows[(size_t)p1.x * a.ld], v2 = drows[(size_t)p2.x * a.ld],
                            v3 = drows[(size_t)p3.x * a.ld];
                acc += __int_as_float(p0.y) * v0; acc += __int_as_float(p1.y) * v1;
                acc += __int_as_float(p2.y) * v2; acc += __int_as_float(p3.y) * v3;
            }
            for (; e < e1; ++e) {
                const int2 pe = ent[e];
                acc += __int_as_float(pe.y) * drows[(size_t)pe.x * a.ld];
            }
        }
        if (a.ginv_off) {
            // the producer stored its rows in list order: the rows of point s are contiguous, ascending grouped row
            const int32_t *off = a.ginv_off + (size_t)b * (a.S + 1) + s0 + g;
            const float *rows = a.gsa + (size_t)b * a.g_rows * a.ld + cc;
            const int e1 = off[1];
            for (int e = off[0]; e < e1; e += 8) {   // absent entries add +0.0f: still the ascending chain
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = e + u < e1 ? rows[(size_t)(e + u) * a.ld] : 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
        }
        if (a.ginv_off2) {   // second scale of the next (MSG) level, same scheme
            const int32_t *off = a.ginv_off2 + (size_t)b * (a.S + 1) + s0 + g;
            const float *rows = a.gsa2 + (size_t)b * a.g_rows2 * a.ld + cc;
            const int e1 = off[1];
            for (int e = off[0]; e < e1; e += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = e + u < e1 ? rows[(size_t)(e + u) * a.ld] : 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
        }
        dsrc[t] = acc;
    }
    __syncthreads();
    // max-pool backward: dZ3[c][g*32+k] = dout[g][c] if k == arg[g][c] else 0.
    // One (point, 8-channel block) per thread; arg/dout reads are broadcasts across the 32 samples.
#pragma unroll
    for (int i = 0; i < NTASK; ++i) {
        const int t = tid + i * NT;
        if (t >= P * nblk || PSG_DIAGBIT(a, 1)) break;
        const int pnt = t % P, blk = t / P;
        const int g = pnt / KS, k = pnt & (KS - 1);
        const uint2 am = am_pre[i];
        const float *dp = dsrc + (size_t)g * a.C3 + blk * 8;
        const float4 d0 = *(const float4 *)dp, d1 = *(const float4 *)(dp + 4);
        float4 v0, v1;
        v0.x = (int)(am.x & 0xFF) == k ? d0.x : 0.f;
        v0.y = (int)((am.x >> 8) & 0xFF) == k ? d0.y : 0.f;
        v0.z = (int)((am.x >> 16) & 0xFF) == k ? d0.z : 0.f;
        v0.w = (int)(am.x >> 24) == k ? d0.w : 0.f;
        v1.x = (int)(am.y & 0xFF) == k ? d1.x : 0.f;
        v1.y = (int)((am.y >> 8) & 0xFF) == k ? d1.y : 0.f;
        v1.z = (int)((am.y >> 16) & 0xFF) == k ? d1.z : 0.f;
        v1.w = (int)(am.y >> 24) == k ? d1.w : 0.f;
        float *dst = buf0 + (size_t)blk * L::BLK + pnt * 8;
        *(float4 *)dst = v0;
        *(float4 *)(dst + 4) = v1;
    }
    __syncthreads();
    if (!PSG_DIAGBIT(a, 8)) layer_bwd<P, NW, MAXT>(a.l3t, buf0, wg);
    if (!PSG_DIAGBIT(a, 16)) __syncthreads();
    if (!PSG_DIAGBIT(a, 8)) layer_bwd<P, NW, MAXT>(a.l2t, buf0, wg);
    if (!PSG_DIAGBIT(a, 16)) __syncthreads();
    if constexpr (P == 32 * NW) {
        if (a.w1c) {   // colour-only request at level 0: three columns of the first layer on the vector pipe (sa_l1t_colour)
            float *orowc = a.gsa_out + (size_t)b * a.S * KS * 4;
            const int32_t *posc = a.gpos_out + (size_t)b * a.S * KS + (size_t)s0 * KS;
            if (a.C1 == 32) sa_l1t_colour<P, NW, 32>(a, buf0, dsrc + a.w1c_off, posc, orowc);
            else sa_l1t_colour<P, NW, 16>(a, buf0, dsrc + a.w1c_off, posc, orowc);
            return;
        }
    }
    if (!a.split) {   // (split: the rows stored below are dZ1 itself; W1f^T is applied per point by pw_bwd_kernel)
        if (!PSG_DIAGBIT(a, 8)) layer_bwd<P, NW, MAXT>(a.l1t, buf0, wg);
        __syncthreads();
    }
    if (PSG_DIAGBIT(a, 32)) return;
    // index_points backward (pointnet_util.py:119,131): the feature rows [c_lo, c_hi) of the grouped-input gradient
    // are stored as plain rows; the consumer (previous level's sa_bwd, or dx0_gather_kernel) sums them through the
    // inverse group lists.  (LDS channel order is [feats, rel_xyz]: LDS channel c is feature channel c.)
    const int nc = a.c_hi - a.c_lo;
    const int32_t *pos = s_pos;
    float *orow = a.gsa_out + (size_t)b * a.S * KS * a.cg_out;
    if (a.cg_out == 4) {   // colour-only request of the attack loop: one 16-byte row {c_lo, c_lo+1, c_lo+2, 0} per lane
        for (int j = tid; j < P; j += NT) {
            const float4 v = make_float4(buf0[L::off(a.c_lo, j)], buf0[L::off(a.c_lo + 1, j)], buf0[L::off(a.c_lo + 2, j)], 0.0f);
            if (pos[j] >= 0) *(float4 *)(orow + (size_t)pos[j] * 4) = v;   // padding rows (always zero) are not listed
        }
        return;
    }
    if (((nc | a.c_lo | a.cg_out) & 3) == 0) {
        // 16 bytes per thread, 32 lanes per row (see fp_bwd_kernel: the element-wise loop below costs a run-time division,
        // a 4-byte LDS read and a 4-byte store per element, 16 rounds per thread at levels 1-3)
        constexpr int RG = NT / 32;
        const int ql = tid & 31, rg = tid >> 5;
        for (int j = rg; j < P; j += RG) {
            const int pj = pos[j];
            if (pj < 0) continue;                                                   // padding rows are not listed
            float *o = orow + (size_t)pj * a.cg_out + a.c_lo;
            for (int q = ql; q < (nc >> 2); q += 32) *(float4 *)(o + 4 * q) = *(const float4 *)(buf0 + L::off(a.c_lo + 4 * q, j));
        }
        return;
    }
    for (int t = tid; t < P * nc; t += NT) {
        const int j = t / nc, c = a.c_lo + (t - j * nc);
        if (pos[j] >= 0) orow[(size_t)pos[j] * a.cg_out + c] = buf0[L::off(c, j)];   // padding rows are not listed
    }
}

// ------------------------------------------------------------------------------------------ FP fwd
// FP split (round 5): the 3-NN interpolation is linear, so the interpolated part of a module's first layer commutes with it,
//     W1 . [f1 ; interp(f2)] = W1a . f1 + interp(W1b . f2),
// and W1b . f2 is a property of the COARSE point: 4 x fewer points than the module has (1024 / 256 / 64 against 4096 / 1024 / 256
// for fp1 / fp2 / fp3).  The coarser module computes T = out . W1b^T as one more layer behind its own last one (FpFwdArgs::extra,
// rows out2); here a point's three rows of T are gathered, weighted, straight into the MFMA ACCUMULATORS of the first layer's
// tiles (lane (j, h) reads the 4 x 16 bytes of each row that its registers hold - the bytes the interpolation read before), and
// the layer's matrix work is the skip columns only: none at all for fp1 (128 of its 128 input channels are interpolated: one
// of the five layers of fp1 + head gone), 64 of 320 for fp2, 128 of 384 for fp3.
template <int P, int NW>
__device__ __forceinline__ void fp_layer1_split(const FpFwdArgs &a, int b, int n0, float *__restrict__ buf, size_t wg_linear)
{
    constexpr int PB = P / 32, BLK = Lds<P>::BLK;
    const FwdLayer &L = a.layer[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ntask = L.mb * PB;
    const int task = (wave + (int)(wg_linear & (NW - 1))) & (NW - 1);
    f32x16 c;
    if (task < ntask) {
        const int mb = task / PB, pb = task - mb * PB;
        const size_t n3 = ((size_t)b * a.N + n0 + pb * 32 + j) * 3;
        const int i0 = a.nn_idx[n3], i1 = a.nn_idx[n3 + 1], i2 = a.nn_idx[n3 + 2];
        const float w0 = a.nn_w[n3], w1 = a.nn_w[n3 + 1], w2 = a.nn_w[n3 + 2];
        const float *tb = a.tsrc + (size_t)b * a.S * a.ldt + mb * 32 + 4 * h;
        const float4 *r0 = (const float4 *)(tb + __umul24((unsigned)i0, (unsigned)a.ldt));
        const float4 *r1 = (const float4 *)(tb + __umul24((unsigned)i1, (unsigned)a.ldt));
        const float4 *r2 = (const float4 *)(tb + __umul24((unsigned)i2, (unsigned)a.ldt));
        const float4 *bp = (const float4 *)(L.bias + mb * 32 + 4 * h);
        // accumulator registers 4g .. 4g+3 of lane (j, h) are channels mb*32 + 8g + 4h + (0..3) of point j.
        // Two halves of six row pieces each, fenced, the bias behind them: with all twelve pieces and the bias in flight the
        // kernel does not fit the 96 VGPRs of five waves per SIMD (see fp_fwd_kernel).
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int g = 2 * half; g < 2 * half + 2; ++g) {
                const float4 u0 = r0[2 * g], u1 = r1[2 * g], u2 = r2[2 * g];
                c[4 * g] = u0.x * w0 + u1.x * w1 + u2.x * w2;
                c[4 * g + 1] = u0.y * w0 + u1.y * w1 + u2.y * w2;
                c[4 * g + 2] = u0.z * w0 + u1.z * w1 + u2.z * w2;
                c[4 * g + 3] = u0.w * w0 + u1.w * w1 + u2.w * w2;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const float4 bq0 = bp[0], bq1 = bp[2], bq2 = bp[4], bq3 = bp[6];
        if (L.k8) c = tile_mac<BLK, false>(L.w + (size_t)mb * L.k8 * 64 + lane, L.k8, buf + (pb * 32 + j) * 8 + 4 * h, c);
        c[0] += bq0.x; c[1] += bq0.y; c[2] += bq0.z; c[3] += bq0.w;
        c[4] += bq1.x; c[5] += bq1.y; c[6] += bq1.z; c[7] += bq1.w;
        c[8] += bq2.x; c[9] += bq2.y; c[10] += bq2.z; c[11] += bq2.w;
        c[12] += bq3.x; c[13] += bq3.y; c[14] += bq3.z; c[15] += bq3.w;
        const unsigned m = relu_bits(c);
        if (L.mask) L.mask[(wg_linear * ntask + task) * 64 + lane] = (uint16_t)m;
    }
    __syncthreads();
    if (task < ntask) {
        const int mb = task / PB, pb = task - mb * PB;
        store_tile<P>(buf, mb, pb * 32 + j, h, c);
    }
}

// BIG: the concatenated input does not fit LDS (MSG fp4: 512 + 1024 channels): the first layer's K is streamed
// through the buffer in chunks of KC blocks, its accumulators staying in registers across the chunks.
template <int P, int NW, bool BIG = false>
// (five waves per SIMD = at most 96 VGPRs: the split first layer and the extra layer took the allocator from 88 to 112 registers,
// i.e. from five resident workgroups per CU to four; with fp_layer1_split's row pieces fenced into two halves it fits 96 unspilled)
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(5))) void fp_fwd_kernel(FpFwdArgs a)
{
    using L = Lds<P>;
    constexpr int NT = NW * 64;
    extern __shared__ float lds[];
    float *buf0 = lds;   // the one activation buffer (layers run in place)
    const int tid = threadIdx.x;
    int bx, b;
    xcd_tile(bx, b);
    const int n0 = bx * P;
    const size_t wg = (size_t)b * gridDim.x + bx;
    if (PSG_DIAGBIT(a, 512) && (tid & 63) == 0) a.dbg[(wg * 8 + (tid >> 6)) * 16 + 15] = __builtin_amdgcn_s_memtime();
    if (PSG_DIAGBIT(a, 256) && tid == 0) {
        a.dbg[wg * 4 + 0] = __builtin_amdgcn_s_memtime();
        a.dbg[wg * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (BIG) {
        static_assert(!BIG || P == 32, "streamed first layer: one 32-point tile per workgroup");
        constexpr int KC = 64;                       // blocks (of 8 channels) per chunk
        constexpr int RG = NT / 32, JI = P / RG;
        const int ql = tid & 31, rg = tid >> 5;
        const int lane = tid & 63, wave = tid >> 6, jj = lane & 31, h = lane >> 5;
        int i0[JI], i1[JI], i2[JI];
        float w0[JI], w1[JI], w2[JI];
#pragma unroll
        for (int u = 0; u < JI; ++u) {
            const size_t n = (size_t)b * a.N + n0 + rg + u * RG;
            i0[u] = a.nn_idx[n * 3]; i1[u] = a.nn_idx[n * 3 + 1]; i2[u] = a.nn_idx[n * 3 + 2];
            w0[u] = a.nn_w[n * 3]; w1[u] = a.nn_w[n * 3 + 1]; w2[u] = a.nn_w[n * 3 + 2];
        }
        const FwdLayer &L0 = a.layer[0];
        const int ntask = L0.mb;                     // P == 32: one tile per 32 output channels
        const int first = (wave + (int)(wg & (NW - 1))) & (NW - 1);
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.0f;
        for (int kb = 0; kb < L0.k8; kb += KC) {
            const int nb = min(KC, L0.k8 - kb), c_lo = kb * 8, c_hi = (kb + nb) * 8;
            // skip-link rows, channels [c_lo, min(c_hi, C1))
#pragma unroll
            for (int u = 0; u < JI; ++u) {
                const int j = rg + u * RG;
                const float4 *f4 = (const float4 *)(a.feat1 + ((size_t)b * a.N + n0 + j) * a.C1);
                for (int q = (c_lo >> 2) + ql; 4 * q < min(c_hi, a.C1); q += 32) *(float4 *)(buf0 + L::off(4 * q - c_lo, j)) = f4[q];
            }
            // interpolated rows, concat channels [max(c_lo, C1), c_hi)
#pragma unroll
            for (int u = 0; u < JI; ++u) {
                const int j = rg + u * RG;
                const float4 *g0 = (const float4 *)(a.feat2 + ((size_t)b * a.S + i0[u]) * a.C2);
                const float4 *g1 = (const float4 *)(a.feat2 + ((size_t)b * a.S + i1[u]) * a.C2);
                const float4 *g2 = (const float4 *)(a.feat2 + ((size_t)b * a.S + i2[u]) * a.C2);
                const int lo = max(c_lo, a.C1) - a.C1, hi = c_hi - a.C1;
                for (int q = (lo >> 2) + ql; 4 * q < hi; q += 32) {
                    const float4 u0 = g0[q], u1 = g1[q], u2 = g2[q];
                    float4 r;
                    r.x = u0.x * w0[u] + u1.x * w1[u] + u2.x * w2[u];
                    r.y = u0.y * w0[u] + u1.y * w1[u] + u2.y * w2[u];
                    r.z = u0.z * w0[u] + u1.z * w1[u] + u2.z * w2[u];
                    r.w = u0.w * w0[u] + u1.w * w1[u] + u2.w * w2[u];
                    *(float4 *)(buf0 + L::off(a.C1 + 4 * q - c_lo, j)) = r;
                }
            }
            __syncthreads();
            if (first < ntask)
                c = tile_mac<L::BLK, false>(L0.w + ((size_t)first * L0.k8 + kb) * 64 + lane, nb, buf0 + jj * 8 + 4 * h, c);
            __syncthreads();
        }
        if (first < ntask) {   // bias, ReLU, mask bits and write-back exactly like layer_fwd
            const float *bp = L0.bias + first * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] += bp[8 * (r >> 2) + (r & 3)];
            const unsigned m = relu_bits(c);
            if (L0.mask) L0.mask[(wg * ntask + first) * 64 + lane] = (uint16_t)m;
            store_tile<P>(buf0, first, jj, h, c);
        }
        __syncthreads();
    } else if (!PSG_DIAGBIT(a, 1)) {
        // 32 consecutive lanes read consecutive float4 of ONE source row (512 contiguous bytes per row group), so a
        // wave-wide load touches 8 cache lines instead of 64; neighbour indices / weights are broadcast loads.
        constexpr int RG = NT / 32, JI = P / RG;
        static_assert(P % RG == 0, "row groups must tile the points");
        const int ql = tid & 31, rg = tid >> 5;
        // address arithmetic: the room's bases are wave-uniform (scalar 64-bit), the per-row offsets are 24-bit products
        // (row < 2^12 .. 2^16, channels < 2^11: v_mul_u32_u24 runs at full rate, a 32-bit v_mul_lo at a quarter of it, and
        // the 64-bit per-thread versions of these cost ~10 % of this kernel's matrix time: vector work is not hidden)
        const int32_t *nn_i = a.nn_idx + ((size_t)b * a.N + n0) * 3;
        const float *nn_wp = a.nn_w + ((size_t)b * a.N + n0) * 3;
        const float *f1b = a.feat1 ? a.feat1 + ((size_t)b * a.N + n0) * a.C1 : nullptr;
        const float *f2b = a.feat2 + (size_t)b * a.S * a.C2;
        int i0[JI], i1[JI], i2[JI];
        float w0[JI], w1[JI], w2[JI];
        if (!a.tsrc) {
#pragma unroll
            for (int u = 0; u < JI; ++u) {
                const int n3 = (rg + u * RG) * 3;
                i0[u] = nn_i[n3]; i1[u] = nn_i[n3 + 1]; i2[u] = nn_i[n3 + 2];
                w0[u] = nn_wp[n3]; w1[u] = nn_wp[n3 + 1]; w2[u] = nn_wp[n3 + 2];
            }
        }
        if (a.feat1) {
#pragma unroll
            for (int u = 0; u < JI; ++u) {
                const int j = rg + u * RG;
                const unsigned o1 = __umul24((unsigned)j, (unsigned)a.C1);
                for (int q = ql; q < (a.C1 >> 2); q += 32) *(float4 *)(buf0 + L::off(4 * q, j)) = *(const float4 *)(f1b + (o1 + 4u * q));
            }
        }
#pragma unroll
        for (int u = 0; u < JI; ++u) {
            if (a.tsrc) break;          // split first layer: the interpolated part arrives through fp_layer1_split
            const int j = rg + u * RG;
            // (uniform base + 32-bit offset: the loads take the SGPR-base form, no 64-bit vector adds per address)
            const unsigned o0 = __umul24((unsigned)i0[u], (unsigned)a.C2), oa = __umul24((unsigned)i1[u], (unsigned)a.C2),
                           ob = __umul24((unsigned)i2[u], (unsigned)a.C2);
            for (int q = ql; q < (a.C2 >> 2); q += 32) {
                const float4 u0 = *(const float4 *)(f2b + (o0 + 4u * q)), u1 = *(const float4 *)(f2b + (oa + 4u * q)),
                             u2 = *(const float4 *)(f2b + (ob + 4u * q));
                float4 r;
                r.x = u0.x * w0[u] + u1.x * w1[u] + u2.x * w2[u];
                r.y = u0.y * w0[u] + u1.y * w1[u] + u2.y * w2[u];
                r.z = u0.z * w0[u] + u1.z * w1[u] + u2.z * w2[u];
                r.w = u0.w * w0[u] + u1.w * w1[u] + u2.w * w2[u];
                *(float4 *)(buf0 + L::off(a.C1 + 4 * q, j)) = r;
            }
        }
    }
    __syncthreads();
    float *in = buf0;
    if (PSG_DIAGBIT(a, 512) && (tid & 63) == 0) a.dbg[(wg * 8 + (tid >> 6)) * 16 + 0] = __builtin_amdgcn_s_memtime();
    for (int l = BIG ? 1 : 0; l < a.n_layers; ++l) {
        if (!BIG && l == 0 && a.tsrc) fp_layer1_split<P, NW>(a, b, n0, in, wg);
        else if (!PSG_DIAGBIT(a, 8)) layer_fwd<P, NW, 1>(a.layer[l], in, wg);
        if (PSG_DIAGBIT(a, 512) && (tid & 63) == 0) a.dbg[(wg * 8 + (tid >> 6)) * 16 + 1 + 2 * l] = __builtin_amdgcn_s_memtime();
        if (!PSG_DIAGBIT(a, 16)) __syncthreads();
        if (PSG_DIAGBIT(a, 512) && (tid & 63) == 0) a.dbg[(wg * 8 + (tid >> 6)) * 16 + 2 + 2 * l] = __builtin_amdgcn_s_memtime();
    }
    // `in` now holds the last layer's output
    if (a.out) {
        if ((a.Cout & 3) == 0) {   // 16 bytes per thread, 32 lanes per row (see fp_bwd_kernel's output rows)
            constexpr int RG = NT / 32;
            const int ql = tid & 31, rg = tid >> 5;
            float *ob = a.out + ((size_t)b * a.N + n0) * a.Cout;
            for (int j = rg; j < P; j += RG) {
                const unsigned o = __umul24((unsigned)j, (unsigned)a.Cout);
                for (int q = ql; q < (a.Cout >> 2); q += 32) *(float4 *)(ob + (o + 4u * q)) = *(const float4 *)(in + L::off(4 * q, j));
            }
        } else {
            for (int t = tid; t < P * a.Cout; t += NT) {
                const int j = t / a.Cout, c = t - j * a.Cout;
                a.out[((size_t)b * a.N + n0 + j) * a.Cout + c] = in[L::off(c, j)];
            }
        }
    }
    if (a.out2) {
        // producer side of the finer module's split first layer (fp_layer1_split): T = out . W1b^T, in place over the output
        // (layer_fwd's barrier sits between the row reads above and its stores)
        layer_fwd<P, NW, 1>(a.extra, in, wg);
        __syncthreads();
        constexpr int RG = NT / 32;
        const int ql = tid & 31, rg = tid >> 5;
        float *ob = a.out2 + ((size_t)b * a.N + n0) * a.Cout2;
        for (int j = rg; j < P; j += RG) {
            const unsigned o = __umul24((unsigned)j, (unsigned)a.Cout2);
            for (int q = ql; q < (a.Cout2 >> 2); q += 32) *(float4 *)(ob + (o + 4u * q)) = *(const float4 *)(in + L::off(4 * q, j));
        }
    }
    if (a.logp && !PSG_DIAGBIT(a, 4)) {
        // log_softmax over the n_cls (<= 16) head rows: 8 lanes per point, classes q and q + 8 per lane
        for (int t = tid; t < P * 8; t += NT) {
            const int j = t >> 3, q = t & 7;
            const bool has0 = q < a.n_cls;
            const float z0 = has0 ? in[L::off(q, j)] : -INFINITY;
            const bool has1 = q + 8 < a.n_cls;
            const float z1 = has1 ? in[L::off(q + 8, j)] : -INFINITY;
            float m = fmaxf(z0, z1);
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            m = fmaxf(m, __shfl_xor(m, 4));
            float s = (has0 ? expf(z0 - m) : 0.0f) + (has1 ? expf(z1 - m) : 0.0f);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            const float lse = logf(s);
            float *o = a.logp + ((size_t)b * a.N + n0 + j) * a.n_cls;
            if (has0) o[q] = (z0 - m) - lse;
            if (has1) o[q + 8] = (z1 - m) - lse;
        }
    }
    if (PSG_DIAGBIT(a, 256) && tid == 0) {
        a.dbg[wg * 4 + 2] = __builtin_amdgcn_s_memtime();
        a.dbg[wg * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------ FP bwd
// Transposed 3-NN interpolation of the finer module, for the 32 points of a workgroup: buf[c][j] = sum over the list of
// point j of w * dint[fine][c] (CSR by coarse point, ascending fine point: the order the sums always had).
// The lists of a room are as uneven as its point density - a coarse point inside a dense patch is the neighbour of hundreds
// of fine points while the mean is 12 - so the rows are not dealt to lanes (one lane walking one list, everybody waiting
// for the longest) but to WAVES by entry count: wave w takes the consecutive rows whose lists start inside the w-th
// NW-th of the workgroup's entry range, walks their entries in order with all 64 lanes on one gradient row (a coalesced
// Cout-float read per entry, eight rows in flight) and writes every row it owns exactly once (zeros for an empty list).
// A wave's share is at most the even share plus one list.  VW = Cg / 64 channels per lane (2 or 4).
template <int P, int NW, int VW>
__device__ __forceinline__ void fp_bwd_gather_rows(const FpBwdArgs &a, int b, int n0, float *__restrict__ buf, int lane, int wave)
{
    static_assert(P == 32, "one 32-point tile per workgroup");
    // gradient rows in flight per wave: the walk is a chain of L2 / MALL round trips (26 % of fp2 backward, 24 % of fp3's: skipped
    // for timing, 5.4 / 2.9 -> 4.0 / 2.2 ms), so as many as the registers allow: 16 two-float pieces, 8 four-float ones
    constexpr int INF = VW <= 2 ? 16 : 8;
    typedef float vwf __attribute__((ext_vector_type(VW)));
    const int32_t *offp = a.nninv_off + (size_t)b * (a.N + 1) + n0;
    const int2 *ent = a.nninv_ent + (size_t)b * 3 * a.n_fine;
    const int offl = offp[lane < 32 ? lane : 32];                       // lanes 32.. hold the end of the range
    const int e0 = __builtin_amdgcn_readfirstlane(offl), e_end = __builtin_amdgcn_readlane(offl, 32);
    const int ltot = e_end - e0;
    const int t_lo = (int)(((long long)ltot * wave) / NW), t_hi = (int)(((long long)ltot * (wave + 1)) / NW);
    const int r_lo = __popcll(__ballot(lane < 32 && offl - e0 < t_lo));
    const int r_hi = wave == NW - 1 ? 32 : __popcll(__ballot(lane < 32 && offl - e0 < t_hi));
    if (r_lo >= r_hi) return;
    const int lo = __builtin_amdgcn_readlane(offl, r_lo), hi = __builtin_amdgcn_readlane(offl, r_hi);
    const float *rows = a.dint + (size_t)b * a.n_fine * a.Cg + lane * VW;
    float *dst = buf + ((lane * VW) >> 3) * Lds<P>::BLK + ((lane * VW) & 7);    // + 8 * point
    int cur = r_lo, next = __builtin_amdgcn_readlane(offl, r_lo + 1);
    vwf acc = 0.0f;
    for (int base = lo; base < hi; base += 64) {
        const int n = hi - base < 64 ? hi - base : 64;
        const int2 pe = lane < n ? ent[base + lane] : make_int2(0, 0);  // (an absent entry reads row 0 with weight 0)
        for (int i = 0; i < n; i += INF) {
            vwf v[INF];
            float w[INF];
#pragma unroll
            for (int k = 0; k < INF; ++k) {
                const int fine = __builtin_amdgcn_readlane(pe.x, i + k);
                w[k] = __int_as_float(__builtin_amdgcn_readlane(pe.y, i + k));
                v[k] = *(const vwf *)(rows + (size_t)fine * a.Cg);
            }
#pragma unroll
            for (int k = 0; k < INF; ++k) {
                if (i + k < n) {
                    const int e = base + i + k;
                    while (e >= next) {                                 // the row is complete (or empty): write it, take the next
                        *(vwf *)(dst + 8 * cur) = acc;
                        acc = 0.0f;
                        ++cur;
                        next = __builtin_amdgcn_readlane(offl, cur + 1);
                    }
                    acc += w[k] * v[k];
                }
            }
        }
    }
    for (; cur < r_hi; ++cur) {
        *(vwf *)(dst + 8 * cur) = acc;
        acc = 0.0f;
    }
}

// BIG: the gradient of the concatenated input does not fit LDS (MSG fp4: 1536 channels): the last transposed layer
// is not run in place; NW output tiles at a time go through a staging area behind its input and out to HBM.
template <int P, int NW, int MAXT, bool BIG = false>
__global__ __launch_bounds__(NW * 64) void fp_bwd_kernel(FpBwdArgs a)
{
    using L = Lds<P>;
    constexpr int PB = P / 32, NT = NW * 64;
    extern __shared__ float lds[];
    float *buf0 = lds;   // the one activation buffer (layers run in place)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, b;
    xcd_tile(bx, b);
    const int n0 = bx * P;
    const size_t wg = (size_t)b * gridDim.x + bx;

    if (a.dout || a.nninv_off) {
        // dZ_last = dout * mask_last, loaded tile-wise so the mask bits line up with the forward epilogue
        const int j = lane & 31, h = lane >> 5;
        const int ntask = a.mb_last * PB;
        // the gathered gradient goes through the activation buffer itself: waves gather whole rows (balanced by entry
        // count), then every lane masks its own tile elements in place
        const bool staged = P == 32 && a.nninv_off && (a.Cg == 128 || a.Cg == 256);
        if constexpr (P == 32) {
            if (staged) {
                if (a.Cg == 128) fp_bwd_gather_rows<P, NW, 2>(a, b, n0, buf0, lane, wave);
                else fp_bwd_gather_rows<P, NW, 4>(a, b, n0, buf0, lane, wave);
                __syncthreads();
            }
        }
        // finer module split: the rows gathered are the gradient of T = out . W1b^T (fp_layer1_split), one transposed layer
        // away from the gradient of this module's output; mask_last is that layer's mask (the host only sets `pre` with a
        // staged gather)
        if (a.pre.w) layer_bwd<P, NW, MAXT>(a.pre, buf0, wg);
        else
        for (int task = wave; task < ntask; task += NW) {
            const int mb = task / PB, pb = task - mb * PB;
            const unsigned m = a.mask_last[(wg * ntask + task) * 64 + lane];
            float4 dq[4];
            if (staged) {
                const float *o = buf0 + (size_t)(mb * 4) * L::BLK + (pb * 32 + j) * 8 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) dq[g] = *(const float4 *)(o + (size_t)g * L::BLK);
            } else if (a.nninv_off) {
                // deterministic transpose of the finer module's interpolation: sum_w * (its gradient rows)
                const int32_t *off = a.nninv_off + (size_t)b * (a.N + 1) + n0 + pb * 32 + j;
                const int2 *ent = a.nninv_ent + (size_t)b * 3 * a.n_fine;
#pragma unroll
                for (int g = 0; g < 4; ++g) dq[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int e1 = off[1];
                // two list entries per pass (both entry records first, then their 8 row pieces): half the dependent
                // memory round trips; an absent second entry contributes w = 0 times row 0
                for (int e = off[0]; e < e1; e += 2) {
                    const int2 pa = ent[e];
                    const int2 pb = e + 1 < e1 ? ent[e + 1] : make_int2(0, 0);
                    const float wa = __int_as_float(pa.y), wb = __int_as_float(pb.y);
                    const float *sa = a.dint + ((size_t)b * a.n_fine + pa.x) * a.Cout + mb * 32 + 4 * h;
                    const float *sb = a.dint + ((size_t)b * a.n_fine + pb.x) * a.Cout + mb * 32 + 4 * h;
                    float4 da[4], db[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) { da[g] = *(const float4 *)(sa + 8 * g); db[g] = *(const float4 *)(sb + 8 * g); }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        dq[g].x += wa * da[g].x; dq[g].y += wa * da[g].y; dq[g].z += wa * da[g].z; dq[g].w += wa * da[g].w;
                    }
                    if (e + 1 < e1) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            dq[g].x += wb * db[g].x; dq[g].y += wb * db[g].y; dq[g].z += wb * db[g].z; dq[g].w += wb * db[g].w;
                        }
                    }
                }
            } else {
                const float *row = a.dout + ((size_t)b * a.N + n0 + pb * 32 + j) * a.Cout + mb * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) dq[g] = *(const float4 *)(row + 8 * g);
            }
            f32x16 v;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 d = dq[g];
                v[4 * g] = ((m >> (4 * g)) & 1u) ? d.x : 0.0f;
                v[4 * g + 1] = ((m >> (4 * g + 1)) & 1u) ? d.y : 0.0f;
                v[4 * g + 2] = ((m >> (4 * g + 2)) & 1u) ? d.z : 0.0f;
                v[4 * g + 3] = ((m >> (4 * g + 3)) & 1u) ? d.w : 0.0f;
            }
            store_tile<P>(buf0, mb, pb * 32 + j, h, v);
        }
    } else {
        // log_softmax backward: dz = dlogp - exp(logp) * sum(dlogp)   (pointnet2_sem_seg.py:38).
        // The workgroup's 32 rows of logp and dlogp are two contiguous runs of 32 x n_cls floats: all threads copy them into
        // LDS (coalesced), then four lanes per point share the row - the first sums dlogp in class order (the order the sum
        // always had), each computes the classes q, q + 4, q + 8, q + 12.  (One thread per point walking its two rows in
        // global memory was a chain of 2 x 13 scattered loads at the head of every workgroup: fp1 backward 10.45 -> 10.1 ms.)
        float *s_lp = buf0 + (size_t)4 * L::BLK, *s_dl = buf0 + (size_t)6 * L::BLK;     // blocks 4 .. 7: free until the first layer stores
        const int nrow = P * a.n_cls;
        const float *lp_g = a.logp + ((size_t)b * a.N + n0) * a.n_cls, *dl_g = a.dlogp + ((size_t)b * a.N + n0) * a.n_cls;
        for (int i = tid; i < nrow; i += NT) { s_lp[i] = lp_g[i]; s_dl[i] = dl_g[i]; }
        __syncthreads();
        for (int t = tid; t < P * 4; t += NT) {
            const int j = t >> 2, q = t & 3;
            const float *lp = s_lp + j * a.n_cls, *dl = s_dl + j * a.n_cls;
            float s = 0.0f;
            if (q == 0)
                for (int c = 0; c < a.n_cls; ++c) s += dl[c];
            s = __shfl(s, (tid & 63) & ~3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = q + 4 * i;
                buf0[L::off(c, j)] = c < a.n_cls ? dl[c] - expf(lp[c]) * s : 0.0f;
            }
        }
        for (int t = tid; t < P * (a.layer[0].k8 - 2); t += NT) {   // K-padding blocks of conv2^T (none for 13 classes)
            float *zp = buf0 + (size_t)(2 + t / P) * L::BLK + (t % P) * 8;
            *(float4 *)zp = make_float4(0.f, 0.f, 0.f, 0.f);
            *(float4 *)(zp + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    float *in = buf0;
    for (int l = 0; l < a.n_layers - (BIG ? 1 : 0); ++l) {
        layer_bwd<P, NW, MAXT>(a.layer[l], in, wg);
        __syncthreads();
    }
    if (BIG) {
        static_assert(!BIG || P == 32, "streamed last layer: one 32-point tile per workgroup");
        const BwdLayer &Lz = a.layer[a.n_layers - 1];   // first forward layer transposed: no mask behind it
        float *stage = buf0 + (size_t)Lz.k8 * L::BLK;   // NW tiles = NW * 4 blocks behind the layer's input
        const int jj = lane & 31, h = lane >> 5;
        for (int t0 = 0; t0 < Lz.mb; t0 += NW) {
            const int mb = t0 + wave;
            if (mb < Lz.mb) {
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.0f;
                c = tile_mac<L::BLK, false>(Lz.w + (size_t)mb * Lz.k8 * 64 + lane, Lz.k8, buf0 + jj * 8 + 4 * h, c);
                store_tile<P>(stage, wave, jj, h, c);
            }
            __syncthreads();
            for (int t = tid; t < P * NW * 32; t += NT) {
                const int j = t / (NW * 32), cc = t - j * (NW * 32), c = t0 * 32 + cc;
                const float v = stage[L::off(cc, j)];
                const size_t n = (size_t)b * a.N + n0 + j;
                if (c < a.C1) a.dfeat1[n * a.C1 + c] = v;
                else if (c < a.C1 + a.C2) a.dint_out[n * a.C2 + (c - a.C1)] = v;
            }
            __syncthreads();
        }
        return;
    }
    // `in` = gradient of the concat input [C1 skip rows | C2 interpolated rows][point], written out as plain rows, 16 bytes per
    // thread: 32 consecutive lanes cover 512 contiguous bytes of one row (no division by a run-time channel count, one
    // ds_read_b128 and one global store per 4 channels; the element-wise loops this replaces cost a quarter-rate integer
    // division and a 4-byte LDS read + store per element: ~600 vector instructions per thread in fp2 backward).
    // The coarser module gathers the interpolated part through the inverse 3-NN lists (pointnet_util.py:308: the transpose
    // of index_points + weighted sum, without atomics).
    // Split first layer (a.split): the layers above stopped at dZ1.  Its rows ARE the interpolated part's gradient as the
    // coarser module wants it (it applies W1b^T per coarse point after its gather: FpBwdArgs::pre); the skip part is
    // W1a^T . dZ1, run after those rows have been read.
    constexpr int RG = NT / 32;
    const int ql = tid & 31, rg = tid >> 5;
    const int w2 = a.split ? a.Cd : a.C2, o2 = a.split ? 0 : a.C1;
    float *d1 = a.dfeat1 ? a.dfeat1 + ((size_t)b * a.N + n0) * a.C1 : nullptr;     // sole writer of the skip-link gradient
    float *d2 = a.dint_out + ((size_t)b * a.N + n0) * w2;
    for (int j = rg; j < P; j += RG) {
        const unsigned o = __umul24((unsigned)j, (unsigned)w2);
        for (int q = ql; q < (w2 >> 2); q += 32) *(float4 *)(d2 + (o + 4u * q)) = *(const float4 *)(in + L::off(o2 + 4 * q, j));
    }
    if (!d1) return;
    if (a.split) {
        layer_bwd<P, NW, MAXT>(a.skipT, in, wg);      // (its barrier separates the row reads above from its stores)
        __syncthreads();
    }
    for (int j = rg; j < P; j += RG) {
        const unsigned o = __umul24((unsigned)j, (unsigned)a.C1);
        for (int q = ql; q < (a.C1 >> 2); q += 32) *(float4 *)(d1 + (o + 4u * q)) = *(const float4 *)(in + L::off(4 * q, j));
    }
}

// ------------------------------------------------------------------------------------------ per-point side of the SA split
// Backward of the split first layer (see sa_fwd_kernel): for the 32 points n of a workgroup
//     dT[n]   = sum of the dZ1 rows of the grouped rows that gathered point n  (the transpose of index_points, as a gather:
//               sa_bwd stored its rows in list order, so they are the CONTIGUOUS slots [off[n], off[n+1]), ascending row)
//     out[n]  = skip[n] + dT[n] . W1f                                            (C1 -> D channels, one MFMA layer)
// `out` is the complete gradient of the level's pooled features: the coarser FP module's skip-link rows plus the transposed
// grouping of the next SA level - what sa_bwd's prologue used to sum per (group, channel) thread by walking the list of its
// point (a wave waited for its longest list).  Here the workgroup's entry range is dealt to the WAVES by entry count, a wave
// walks its rows with all 64 lanes on one row (coalesced C1-float reads, eight in flight), exactly like fp_bwd_gather_rows.
struct PwBwdArgs {
    const int32_t *ginv_off;   // [B][N + 1]
    const float *gsa;          // [B][g_rows][C1] dZ1 rows in list order
    int g_rows;
    const float *skip;         // [B][N][D] or null
    float *out;                // [B][N][D]
    BwdLayer wt;               // W1f transposed: k8 = C1 / 8, mb = D / 32, no mask
    int N, C1, D;
};

template <int P, int NW, int VW>
__device__ __forceinline__ void pw_bwd_gather_rows(const PwBwdArgs &a, int b, int n0, float *__restrict__ buf, int lane, int wave)
{
    static_assert(P == 32, "one 32-point tile per workgroup");
    typedef float vwf __attribute__((ext_vector_type(VW)));
    const int32_t *offp = a.ginv_off + (size_t)b * (a.N + 1) + n0;
    const int offl = offp[lane < 32 ? lane : 32];                       // lanes 32.. hold the end of the range
    const int e0 = __builtin_amdgcn_readfirstlane(offl), e_end = __builtin_amdgcn_readlane(offl, 32);
    const int ltot = e_end - e0;
    const int t_lo = (int)(((long long)ltot * wave) / NW), t_hi = (int)(((long long)ltot * (wave + 1)) / NW);
    const int r_lo = __popcll(__ballot(lane < 32 && offl - e0 < t_lo));
    const int r_hi = wave == NW - 1 ? 32 : __popcll(__ballot(lane < 32 && offl - e0 < t_hi));
    if (r_lo >= r_hi) return;
    const int lo = __builtin_amdgcn_readlane(offl, r_lo), hi = __builtin_amdgcn_readlane(offl, r_hi);
    const float *rows = a.gsa + (size_t)b * a.g_rows * a.C1 + lane * VW;
    float *dst = buf + ((lane * VW) >> 3) * Lds<P>::BLK + ((lane * VW) & 7);    // + 8 * point
    int cur = r_lo, next = __builtin_amdgcn_readlane(offl, r_lo + 1);
    vwf acc = 0.0f;
    for (int base = lo; base < hi; base += 8) {
        vwf v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = base + k < hi ? *(const vwf *)(rows + (size_t)(base + k) * a.C1) : (vwf)0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = base + k;
            if (e < hi) {
                while (e >= next) {                                     // the point's list is complete (or empty): write, take the next
                    *(vwf *)(dst + 8 * cur) = acc;
                    acc = 0.0f;
                    ++cur;
                    next = __builtin_amdgcn_readlane(offl, cur + 1);
                }
                acc += v[k];
            }
        }
    }
    for (; cur < r_hi; ++cur) {
        *(vwf *)(dst + 8 * cur) = acc;
        acc = 0.0f;
    }
}

template <int P, int NW, int MAXT = 1>
__global__ __launch_bounds__(NW * 64) void pw_bwd_kernel(PwBwdArgs a)
{
    using L = Lds<P>;
    constexpr int NT = NW * 64;
    extern __shared__ float lds[];
    float *buf0 = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, b;
    xcd_tile(bx, b);
    const int n0 = bx * P;
    const size_t wg = (size_t)b * gridDim.x + bx;
    if (a.C1 == 64) pw_bwd_gather_rows<P, NW, 1>(a, b, n0, buf0, lane, wave);
    else if (a.C1 == 128) pw_bwd_gather_rows<P, NW, 2>(a, b, n0, buf0, lane, wave);
    else pw_bwd_gather_rows<P, NW, 4>(a, b, n0, buf0, lane, wave);
    __syncthreads();
    // the skip-link rows this thread will add at the end: issued now, so the round trip runs under the layer (fetched inside the
    // store loop below they were a chain of misses at the tail of the workgroup); up to 8 pieces per thread (D <= 512)
    constexpr int RG = NT / 32, ROWS = P / RG, MAXQ = 4;
    const int ql = tid & 31, rg = tid >> 5;
    const float *sb = a.skip ? a.skip + ((size_t)b * a.N + n0) * a.D : nullptr;
    float4 sk[ROWS][MAXQ];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int q = ql + 32 * i;
            sk[r][i] = (sb && q < (a.D >> 2)) ? *(const float4 *)(sb + (__umul24((unsigned)(rg + r * RG), (unsigned)a.D) + 4u * q))
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    layer_bwd<P, NW, MAXT>(a.wt, buf0, wg);
    __syncthreads();
    {
        float *ob = a.out + ((size_t)b * a.N + n0) * a.D;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int j = rg + r * RG;
            const unsigned o = __umul24((unsigned)j, (unsigned)a.D);
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) {
                const int q = ql + 32 * i;
                if (q < (a.D >> 2)) {
                    float4 v = *(const float4 *)(buf0 + L::off(4 * q, j));
                    const float4 s = sk[r][i];
                    v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
                    *(float4 *)(ob + (o + 4u * q)) = v;
                }
            }
        }
    }
}

}  // namespace psg
