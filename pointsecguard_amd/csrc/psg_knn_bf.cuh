// Dilated kNN graph in feature space (C = 64) with a bf16 PREFILTER on the matrix cores: the exact fp32 distance - the one
// that decides the reference's neighbour order, ResGCN/gcn_lib/dense/torch_edge.py:32-59 - is evaluated only for the few
// candidates per row that can hold one of the wanted ranks; everything else is decided on a LOWER BOUND of the distance
// with a proven, per-pair error term.  Same results, bit for bit, as the exact kernel (psg_knn_fused.cuh), which stays in
// the same launch as the fallback for every tile whose bound cannot be kept.
//
// Approximation.  x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|.  One 32 x 32 tile of
//     acc = (hi_i.lo_j + lo_i.hi_j) + hi_i.hi_j - (1 - B) |x_j|^2 / 2 + (A / 2) |x_i||x_j|
// is 13 v_mfma_f32_32x32x16_bf16 (K = 3 x 64 + one augmented step): 3/16 of the fp32 MFMA time of the exact kernel.
// L = (1 - G) |x_i|^2 - 2 acc = D~ - e_ij with D~ the approximate distance and e_ij = A |x_i||x_j| + B |x_j|^2 + G |x_i|^2.
// Error of D~ against the reference's fp32 distance D (psg_knn_ops.cuh builds the operands): dropped products (lo.lo and
// the split residues) <= 3.02 * 2^-18 |x_i||x_j|; fp32 accumulation, every one of an instruction's 17 additions rounded to
// nearest against the partial sums it can meet: the chain runs the eight SMALL products first (round 6; partial sums <=
// 2^-8 |x_i||x_j|: 8 * 17 * 2^-24 * 2^-8), then the four hi.hi steps (4 * 17 * 2^-24 (1 + 2^-8) |x_i||x_j|), the augmented step
// last (17 * 2^-24 (|x_i||x_j| + |x_j|^2 / 2)): <= 86 * 2^-24 |x_i||x_j| + 8.5 * 2^-24 |x_j|^2 (rounds 3-5 interleaved hi.hi, hi.lo,
// lo.hi per k-step, so all 13 instructions met full-size partial sums, and the |x_j|^2 term all 13: 221 * 2^-24 of both); the
// reference's own chain 64 * 2^-24 |x_i||x_j| and its two distance roundings <= 2^-23 (|x_i|^2 + |x_j|^2) + 2^-22 |x_i||x_j|:
// in distance units (dot errors count twice) |D~ - D| <= 2^-14.57 |x_i||x_j| + 2^-19.75 |x_j|^2 + 2^-23 |x_i|^2.  The kernel
// uses A = 2^-14, B = 2^-19, G = 2^-20 (1.48 x, 1.7 x, 8 x that bound; tools/bf16_split_probe.hip measured 2^-16 of the
// largest |x|^2 on hardware over eight data sets: random, offset, 1e6 dynamic range, near-cancelling pairs, 1e-15 and 1e15
// magnitudes; round 6, per PAIR against these constants: |D~ - D| <= 0.19 - 0.56 e_ij over 8 x 1M pairs, the worst set the one
// with per-channel scales 1e-3..1e3), so L <= D <= L + 2 e_ij - and that is CHECKED for every finalist (kb_final_rows).  Rounds 3-5 ran with
// A = 2^-13.5, B = 2^-15.5: the row margin E_i, and with it the finalists per wanted rank, is ~40 % smaller now.
// The error scales with the PAIR's norms, not with the room's largest norm: the network's own features have a few points
// whose |x|^2 is 40 x the mean, and a bound in units of that maximum was wider than the neighbour distances themselves.
//
// Row margin.  For every candidate j with L_ij <= T (a threshold in distance units): (|x_j| - |x_i|)^2 <= D <= T + 2 e_ij,
// hence |x_j| <= 1.0148 |x_i| + 1.015 sqrt(T) and 2 e_ij <= E_i(T) = 2 (A |x_i| t + B t^2 + G |x_i|^2) at that t: one
// number per row and threshold bounds the gap between L and D for everything the row can hold.
//
// Keys.  key = round((L + 32 u_i) / u_i) with the row's own unit u_i = E_i(|x_i|^2) / 16 ~ 2^-14.9 |x_i|^2 (one FMA whose
// constant carries 2^23: the key is the result's mantissa); a row entry is key << 12 | index (32 bits: twice the rows of
// the exact kernel in the same LDS).  Candidates farther than 2^20 units (35 |x_i|^2) are never kept; a row that
// needs them takes the exact path.
//
// Threshold from a SAMPLE (round 4; rounds 2-3 cut the rows on a growth schedule, 1-3 histogram cuts of 512-1024
// entries per row, each a workgroup-wide stop).  Wave w streams the candidate tiles w S, w S + 1, .. (S steps), so step
// 0 of the 16 waves is a STRIDED sample of the room: up to 512 candidates spread over it.  Step 0 stores every pair's
// entry (fixed slots, no atomics); ONE cut then finds keyX with #(sample keys <= keyX) >= r and sets the row's
// threshold for the rest of the stream, tau = keyX + E_i(keyX) / u_i.  r is chosen on the host so that, for candidates
// in random order, fewer than 1 row in 10^7 ends with less than KK keys <= keyX (the hypergeometric tail: the row
// falls short iff the sample holds r of the KK - 1 nearest); the row then holds ~N r / 512 entries (1.6 KK at d = 27,
// ~100 at d = 1) and is never cut again - no barrier until the stream ends.  EXACTNESS does not rest on the sample: the
// final phase COUNTS the entries with key <= keyX; if there are at least KK, at least KK candidates have D <= keyX u +
// E_i, so the true KK-th distance is below that and every candidate that matters has L below it, i.e. is in the row.  A
// row with fewer (candidates sorted by distance, a sample that missed), or more than CAP entries, sends its tile to the
// exact path: adversarial orders cost time, never correctness (tests/test_gpu_knn_bf16.py).
//
// Final ranks.  Wanted: exact ranks 0, d, .., (k-1) d.  With l_m the m-th smallest L of the row, l_m <= D_(m) <= l_m + E,
// so the element of exact rank m has its L within E of l_m; everything with L below l_m - E is smaller than it,
// everything above l_m + E larger.  Entries are binned by key (256 or 1024 bins of a power-of-two width: exact integer
// edges), the bin that holds approximate rank m and w = ceil(E / width) bins to each side are flagged, overlapping
// windows merge into RUNS, the members of flagged bins ("finalists", a few per wanted rank) get their exact distance
// (ascending-k fmaf chain from the fp32 copy, the exact kernel's arithmetic) and rank themselves inside their run: first
// rank of the run + the number of smaller finalists of the same run IS the exact rank.  More than 256 finalists in a
// row send the tile to the exact path.
//
// The cut and the final ranking run one row per HALF wave (the wave's rows wave and wave + 16 side by side, every wave
// of the workgroup at the same time): per-row reductions are 16-lane DPP steps plus one cross-row exchange and serve
// both rows with one instruction stream; what is left is per-entry work, 32 lanes per row, over as many registers as the
// longer of the two rows needs (these phases are bound by vector instruction issue - 4 cycles per wave instruction and
// SIMD -, so they are written for instruction count: no sentinel work beyond the rows' lengths, 256 bins for short rows).
#pragma once

constexpr int KB_WAVES = 16;
constexpr int KB_Q = 32;                      // query rows per workgroup (one 32-column MFMA tile)
constexpr int KB_CAP = 1024;                  // entries per row buffer
constexpr int KB_ROW = KB_CAP + 2;            // dwords between rows: 8-byte aligned, skewed over the LDS banks
constexpr float KB_A = KNN_BF_A, KB_B = KNN_BF_B, KB_G = KNN_BF_G;   // 2^-14, 2^-19, 2^-20 (psg_knn_ops.cuh)
// (round 4: units of E / 16 instead of E / 8 - the +- 3 units of key quantisation in every margin were a third of it: 20 %
// fewer finalists at d >= 9; E / 32 gave the same counts, the windows are then limited by the bin width)
constexpr int KB_UNIT_DIV = 16;               // the key unit is E_i(|x_i|^2) / KB_UNIT_DIV
constexpr float KB_OFF = 32.0f;               // key offset in units: L of the query itself (>= -2 e_ii ~ -6.4 u) stays positive
constexpr unsigned KB_KEYMAX = 0xFFFFEu;      // key of a candidate beyond the key range (never kept)
constexpr unsigned KB_TAUMAX = 0xFFFF0u;      // largest row threshold (admission lets keys up to tau + 3 through: they must fit 20 bits)
constexpr int KB_MAXFIN = 256;
#ifndef KB_LONG_BINS
#define KB_LONG_BINS 2048                     // bins of the final ranking for rows of more than 256 entries (A/B builds: 1024)
#endif
constexpr int KB_SCR_BITMAP = 64;             // dwords of flagged-bin bitmap (2048 bins)
constexpr int KB_SCR = 64 + KB_SCR_BITMAP + 32;   // dwords of scratch per row in the final ranking: 64 query features, the bitmap, 32 run table
constexpr int KB_RP = 8;                      // floats of row parameters: |x|^2, |x|, unit, 1 / unit, key constant, spare x3

typedef __bf16 kb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float kb_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned kb_u32x4 __attribute__((ext_vector_type(4)));     // one 16-byte operand fragment

// Diagnostic build (EXTRA=-DPSG_KF_TL, tools/knn_timeline.py): thread 0 adds the time since kernel start at the workgroup's
// barrier exits into the exact kernel's g_kf_tl table (slots: 0 start, 1 sample step stored, 2 cut done, 8 stream end,
// 9 wave 0 ranked, 10 all ranked)
#ifdef PSG_KF_TL
#define KB_TL(idx) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    atomicAdd(&g_kf_tl[(idx)], t_ - tl0); atomicAdd(&g_kf_tl[16 + (idx)], 1ull); } } while (0)
__device__ unsigned long long g_kb_ph[32];      // phase cycle sums of the final ranking / cut (wave 0 of every workgroup)
#else
#define KB_TL(idx)
#endif
// (third build, -DPSG_KF_TL -DPSG_KF_PH: stamps inside the cut / final ranking of wave 0; they wait for their own global
// atomics, so read the shares of the phases, not the lengths, and take the barrier timeline from a build without them)
#if defined(PSG_KF_TL) && defined(PSG_KF_PH)
#define KB_PH_BEGIN unsigned long long ph_t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_t0_) :: "memory")
#define KB_PH(idx) do { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (threadIdx.x == 0) { atomicAdd(&g_kb_ph[(idx)], t_ - ph_t0_); atomicAdd(&g_kb_ph[16 + (idx)], 1ull); } ph_t0_ = t_; } while (0)
#else
#define KB_PH_BEGIN
#define KB_PH(idx)
#endif

struct KnnBfArgs {
    const kb_u32x4 *bp;   // [rows / 32][9][64] fragments (psg_knn_ops.cuh)
    const float *sq;      // [rows]
    const float *x;       // [rows][ld] the fp32 features, row-major (finalists' exact distances)
    int ld;
    int32_t *out;         // [rows][k]
    int N;                // points per room, multiple of 32
    int k, d, KK;
    unsigned magic;       // ceil(2^18 / d)
    unsigned rsel;        // the cut after step 0 keeps the sample keys up to the rsel-th smallest (+ the row margin)
    int nsamp;            // sample tiles per wave: 1, or 2 for long rows (N a multiple of 512, at least 4 steps)
    int fine_cut;         // second histogram level inside the threshold's bin (long rows: 20 % fewer entries, no overflows up to d = 20)
    KnnFusedArgs exact;            // the exact path's arguments (xp, sq, out, ...)
    unsigned long long *stats;     // optional [8]: tiles, fallback tiles, rows, finalists, rows cut, entries at the end, and the reasons
                                   // of the fallback tiles in 16-bit fields: [6] no threshold | row above CAP | row short of KK
                                   // entries, [7] row short of KK keys below keyX | more than 256 finalists | bound broken
};

// ---- helpers on the two 32-lane halves of a wave (a row per half)
__device__ __forceinline__ unsigned kb_half_max_u32(unsigned v)          // result in every lane of the half
{
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__shfl_xor((int)v, 16); v = o > v ? o : v;
    return v;
}
__device__ __forceinline__ unsigned kb_half_sum_u32(unsigned v)          // result in every lane of the half
{
    v += (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false);
    v += (unsigned)__shfl_xor((int)v, 16);
    return v;
}
__device__ __forceinline__ unsigned kb_half_incl_scan_u32(unsigned v)    // inclusive prefix sum inside each half
{
    v += dpp_get<0x111, 0xF>(v);   // row_shr:1
    v += dpp_get<0x112, 0xF>(v);
    v += dpp_get<0x114, 0xF>(v);
    v += dpp_get<0x118, 0xF>(v);   // inclusive scan inside each row of 16 lanes
    v += dpp_get<0x142, 0xA>(v);   // row_bcast15: the upper row of each half += total of the lower one
    return v;
}
// number of set bits of the lane's half of `ballot` below the lane
__device__ __forceinline__ unsigned kb_half_prefix(unsigned long long ballot, int h)
{
    const unsigned lo = __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0u), hi = __builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32), 0u);
    return h ? hi : lo;
}
__device__ __forceinline__ unsigned kb_half_count(unsigned long long ballot, int h)
{
    return (unsigned)__popc(h ? (unsigned)(ballot >> 32) : (unsigned)ballot);
}

// the value of quad lane g ^ 1 / g ^ 2 (quad_perm [1,0,3,2] / [2,3,0,1])
__device__ __forceinline__ float kb_quad_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float kb_quad_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); }

// E_i(T) / u_i: the gap between L and D, in key units, for every candidate of the row with L <= T (see the header)
__device__ __forceinline__ unsigned kb_margin_keys(float a, float sq, float inv_unit, float T)
{
    const float t = 1.0148f * a + 1.015f * sqrtf(fmaxf(T, 0.0f));
    const float E = 2.0f * (KB_A * a * t + KB_B * t * t + KB_G * sq);
    const float m = E * inv_unit * 1.0001f + 2.0f;        // + quantisation of the two keys compared
    return m < 1.0e6f ? (unsigned)m + 1u : 1000000u;
}

// ---- the cut and the final ranking: the wave's two rows side by side, one per half wave; lane l of a half holds the
// entries 32 i + l of its row; the row's own LDS is scratch once the entries are in registers.  LDS locations travel as
// dword offsets into the workgroup's dynamic LDS, so that every access stays a ds_ instruction.
struct KbRowArgs {
    unsigned row_dw;        // the first row's entries; the second row's are 16 rows further
    unsigned rp_dw;         // row parameters [KB_Q][KB_RP] floats
    unsigned cnt_dw, thr_dw, tau_dw, keyx_dw;   // per-row counters, admission thresholds on acc, key thresholds, keyX of the cut
    unsigned T[2];          // entries held
    unsigned KK;
    int wave;
};

// The cut after step 0: nv0 full registers of sample entries per lane (32 nv0 per row, the same for every row).  256 bins
// of a power-of-two width over [smallest key, largest key]; the bin in which the cumulative count reaches rsel gives
// keyX = that bin's upper edge (exact integers); tau = keyX + E_i(keyX) / u_i; every entry with key <= tau is kept,
// compacted in place.  Lane 0 of each half stores the row's count and thresholds; returns true (in every lane of the
// half) when the row cannot go on (fewer than rsel sample keys in range, or a threshold beyond the key range).
template <int NVC>
__device__ __forceinline__ bool kb_cut_sample(const KbRowArgs c, int lane, int nv0, unsigned rsel, bool fine)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    const int h = lane >> 5, l5 = lane & 31;
    const int ridx = c.wave + KB_WAVES * h;
    unsigned *row = (unsigned *)kb_smem + c.row_dw + h * KB_WAVES * KB_ROW;
    const float *rp = (const float *)kb_smem + c.rp_dw + ridx * KB_RP;
    unsigned v[NVC];
    unsigned mx = 0u, mnn = 0u;                             // mnn = max of ~v = ~min v
    KB_PH_BEGIN;
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
        v[i] = 0xFFFFFFFFu;
        if (i < nv0) {
            v[i] = row[i * 32 + l5];
            mx = v[i] > mx ? v[i] : mx;
            mnn = ~v[i] > mnn ? ~v[i] : mnn;
        }
    }
    const float sq = rp[0], a = rp[1], unit = rp[2], inv_unit = rp[3], ci = rp[4];
    wave_lds_fence();
    KB_PH(0);
    const unsigned kmx = kb_half_max_u32(mx) >> 12, kmn = (~kb_half_max_u32(mnn)) >> 12;
    const unsigned range = kmx - kmn;
    unsigned sh = range > 253u ? (unsigned)(32 - __builtin_clz(range)) - 8u : 0u;
    if ((range >> sh) > 253u) ++sh;                         // bins 0 .. range >> sh <= 253
    const unsigned base = kmn << 12, s2 = 12u + sh;
    // 256 counters per row; the lane owns bins 8 l .. 8 l + 7
#pragma unroll
    for (int j = 0; j < 4; ++j) ((uint2 *)row)[4 * l5 + j] = make_uint2(0u, 0u);
    wave_lds_fence();
    KB_PH(1);
#pragma unroll
    for (int i = 0; i < NVC; ++i)
        if (i < nv0) atomicAdd(&row[(v[i] - base) >> s2], 1u);
    wave_lds_fence();
    KB_PH(2);
    // the bin in which the cumulative count of the 256 counters reaches `target`, and the entries below that bin
    unsigned ownh = 0u;
    auto find_bin = [&](unsigned target, unsigned &below_out) -> unsigned {
        unsigned cw[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint2 t = ((const uint2 *)row)[4 * l5 + j]; cw[2 * j] = t.x; cw[2 * j + 1] = t.y; }
        unsigned tot = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) tot += cw[j];
        const unsigned incl = kb_half_incl_scan_u32(tot), start = incl - tot;
        unsigned mybin = (unsigned)(8 * l5), s = start, sb = start;
#pragma unroll
        for (int j = 0; j < 7; ++j) { s += cw[j]; const bool lt = s < target; mybin += lt ? 1u : 0u; sb = lt ? s : sb; }
        const unsigned long long own = __ballot(start < target && target <= incl);       // at most one lane per half
        ownh = h ? (unsigned)(own >> 32) : (unsigned)own;
        const int owner = (ownh ? __builtin_ctz(ownh) : 0) + 32 * h;
        const unsigned pk = (unsigned)__shfl((int)(mybin | (sb << 16)), owner);
        below_out = pk >> 16;
        return pk & 0xFFFFu;
    };
    unsigned below1 = 0u;
    const unsigned binB = find_bin(rsel, below1);
    const bool none = ownh == 0u;
    // second level inside bin binB (bins wider than one key): 2^(sh - sh2) sub-bins of width 2^sh2 (one key when sh <= 8), so
    // that keyX is the rsel-th smallest sample key itself, not the edge of a bin that may hold a dozen of them
    unsigned keyX = kmn + ((binB + 1u) << sh) - 1u;
    if (fine && __ballot(sh != 0u) != 0ull) {
        const unsigned sh2 = sh > 8u ? sh - 8u : 0u, smask = (1u << (sh - sh2)) - 1u;
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 4; ++j) ((uint2 *)row)[4 * l5 + j] = make_uint2(0u, 0u);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < NVC; ++i)
            if (i < nv0) {
                if (((v[i] - base) >> s2) == binB) atomicAdd(&row[((v[i] - base) >> (12u + sh2)) & smask], 1u);
            }
        wave_lds_fence();
        unsigned below2 = 0u;
        const unsigned subB = find_bin(rsel - below1, below2);
        keyX = kmn + (binB << sh) + ((subB + 1u) << sh2) - 1u;
    }
    const unsigned margin = kb_margin_keys(a, sq, inv_unit, ((float)keyX - KB_OFF) * unit);
    const unsigned tauk = keyX + margin;
    const bool bad = none || ownh == 0u || keyX >= KB_TAUMAX || tauk > KB_TAUMAX;
    const unsigned keepmax = bad ? 0u : ((tauk << 12) | 0xFFFu);
    KB_PH(3);
    unsigned cb = 0;
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
        if (i < nv0) {
            const bool keep = v[i] <= keepmax;
            const unsigned long long bl = __ballot(keep);
            if (keep) row[cb + kb_half_prefix(bl, h)] = v[i];
            cb += kb_half_count(bl, h);
        }
    }
    if (l5 == 0) {
        ((unsigned *)kb_smem)[c.cnt_dw + ridx] = cb;
        ((unsigned *)kb_smem)[c.tau_dw + ridx] = tauk;
        ((unsigned *)kb_smem)[c.keyx_dw + ridx] = keyX;
        // admit iff key <= tauk, i.e. acc * m2 + ci < tauk + 1; two more units cover the float evaluation
        ((float *)kb_smem)[c.thr_dw + ridx] = (ci - (float)(tauk + 3u)) * (0.5f * unit);
    }
    KB_PH(4);
    return bad;
}

// Final ranking (see the header).  Row LDS: NB bins of 16 bits (counts, then the entries below each bin) in dwords
// [0, NB / 2) - the whole row buffer at NB = 2048 -; once the windows are flagged the bins are dead and the same words hold
// the finalists' exact composites grouped by run (dwords [0, 512)) and the finalist list (entries, dwords [512, 768)).
// scr = KB_SCR words of wave-private LDS per row: the query's 64 features, up to 64 words of flagged-bin bitmap (whose
// first 16 words count the finalists placed per run afterwards), the run table {first bin | first rank << 16, first
// finalist slot | finalists << 16}.  nv = the registers per lane the longer of the two rows needs (<= NV).
// Returns the number of finalists of the two rows, or 0xFFFFFFFF when a row must take the exact path.
//
// Bins.  A window is the bin of the wanted rank and w = ceil(margin / bin width) bins to each side, and with 1024 bins over
// a long row a bin was 2 - 4 margins wide (w = 1: three bins where +- one margin would do) - the finalists were a matter of
// bin width, not of the bound (round 6, tests/golden/gcn28_room.npz features: 88 finalists per row at d = 20 with 1024 bins,
// 55 with 2048, 42 with windows of exactly +- one margin; rounds 4-5 had 1024).  Rows of more than 256 entries now take
// 2048 bins; the finalist list dropped its per-finalist position in bin order (a returning LDS atomic on the bin) to fit.
//
// Runs.  The windows of the wanted ranks are ordered; window j opens a new run when it neither overlaps nor touches window
// j - 1.  A run is a stretch of bins; EVERY entry of its bins is a finalist, so the finalists of run R are exactly the
// entries of ranks-by-bin [cs_R, ce_R) (cs = entries below the run's first bin).  The exact pass finds a finalist's run
// from its bin and takes the next free slot of the run's stretch [fs_R, fs_R + len_R) of the composite array (fs =
// finalists of earlier runs): finalists end up grouped by run, and a finalist ranks itself against its own run only:
// exact rank = cs_R + the number of smaller composites among the run's len_R (a run is a handful of finalists per wanted
// rank it covers; rounds 2-3 compared every finalist with every other of the row: 36 instructions per pair).
struct KbFinalArgs {
    const float *x, *sq;      // row-major features [rows][ld], squared norms
    unsigned ld;
    int32_t *out;             // the first row's output; the second row's is 16 k further
    size_t room_row0, qglob;  // global row of the room's first point and of the first row's query (second: + 16)
    unsigned scr_dw, k, d, magic;
};

template <int NV, int NB>
__device__ __forceinline__ unsigned kb_final_rows(const KbRowArgs c, const KbFinalArgs a, int lane, int nv)
{
    constexpr int LOGNB = NB == 2048 ? 11 : (NB == 1024 ? 10 : 8);
    static_assert(NB == 2048 || NB == 1024 || NB == 256, "bins");
    static_assert(NB / 32 <= KB_SCR_BITMAP, "bitmap words");
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    const int h = lane >> 5, l5 = lane & 31;
    const int ridx = c.wave + KB_WAVES * h;
    unsigned *row = (unsigned *)kb_smem + c.row_dw + h * KB_WAVES * KB_ROW;
    const float *rp = (const float *)kb_smem + c.rp_dw + ridx * KB_RP;
    float *qbuf = (float *)kb_smem + a.scr_dw + h * KB_SCR;
    unsigned *bitmap = (unsigned *)kb_smem + a.scr_dw + h * KB_SCR + 64;
    unsigned *rcnt = bitmap;                                                       // [16] finalists placed per run (after the collect pass)
    uint2 *rt = (uint2 *)((unsigned *)kb_smem + a.scr_dw + h * KB_SCR + 64 + KB_SCR_BITMAP);   // [16] {first bin | cs << 16, fs | len << 16}
    const unsigned T = h ? c.T[1] : c.T[0];
#if defined(KB_ABLATE) && KB_ABLATE == 3      // (timing experiments only, tools/knn_ablate.py: results are wrong)
    return T & 1u;
#endif
    const unsigned tl5 = T > (unsigned)l5 ? T - (unsigned)l5 : 0u;      // entry 32 i + l5 exists iff 32 i < tl5
    const unsigned keyx_ent = (((const unsigned *)kb_smem)[c.keyx_dw + ridx] << 12) | 0xFFFu;
    const unsigned tau_row = ((const unsigned *)kb_smem)[c.tau_dw + ridx];
    unsigned v[NV];
    unsigned mx = 0u, mnn = 0u, nle = 0u;
    KB_PH_BEGIN;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = 0xFFFFFFFFu;
        if (i < nv) {
            const unsigned x = row[i * 32 + l5];
            const bool valid = (unsigned)(32 * i) < tl5;
            v[i] = valid ? x : 0xFFFFFFFFu;
            mx = valid && x > mx ? x : mx;
            mnn = ~v[i] > mnn ? ~v[i] : mnn;
            nle += v[i] <= keyx_ent ? 1u : 0u;
        }
    }
    const float sq = rp[0], aa = rp[1], unit = rp[2], inv_unit = rp[3];
    // the query's 64 features for the exact chains
    const size_t qg = a.qglob + (size_t)(h * KB_WAVES);
    {
        const float *xq = a.x + qg * a.ld;
        qbuf[l5] = xq[l5];
        qbuf[l5 + 32] = xq[l5 + 32];
    }
#pragma unroll
    for (int j = 0; j < NB / 1024; ++j) bitmap[l5 + 32 * j] = 0u;
    if (NB < 1024 && l5 < NB / 32) bitmap[l5] = 0u;
    if (l5 < 16) rt[l5] = make_uint2(0xFFFFFFFFu, 0u);             // (runs the row does not have: their first bin is beyond every bin)
    wave_lds_fence();                                              // (the entries are in registers: their words become the bins)
    // packed 16-bit counters
#pragma unroll
    for (int j = 0; j < NB / 256; ++j) ((uint4 *)row)[32 * j + l5] = make_uint4(0u, 0u, 0u, 0u);
    wave_lds_fence();
    KB_PH(5);
    const unsigned kmx = kb_half_max_u32(mx) >> 12, kmn = (~kb_half_max_u32(mnn)) >> 12;
    // the exactness test (see the header): at least KK entries at or below the cut's keyX
    const bool short_row = kb_half_sum_u32(nle) < c.KK;
    if (__ballot(short_row) != 0ull) return 0xFFFFFFFFu;           // (codes: FFFFFFFF short row, ..FE finalists, ..FD bound)
    // the gap between L and D for everything the row holds: from its threshold, or from its largest key
    const unsigned ktop = tau_row < kmx ? tau_row : kmx;
    const unsigned margin = kb_margin_keys(aa, sq, inv_unit, ((float)ktop - KB_OFF) * unit);
    const unsigned range = kmx - kmn;
    unsigned sh = range > (unsigned)(NB - 2) ? (unsigned)(32 - __builtin_clz(range)) - (unsigned)LOGNB : 0u;
    if ((range >> sh) > (unsigned)(NB - 2)) ++sh;                   // bins 0 .. range >> sh <= NB - 2
    const unsigned w = (margin + (1u << sh) - 1u) >> sh;            // window half width in bins
    const unsigned base = kmn << 12, s2 = 12u + sh;
    unsigned short *cum16 = (unsigned short *)row;
    unsigned long long *sorted = (unsigned long long *)row;         // [256] composites grouped by run (after the collect pass)
    unsigned *fin = row + 512;                                      // [256] entries of the finalists
    KB_PH(6);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i < nv) {
            if ((unsigned)(32 * i) < tl5) {
                const unsigned bb = (v[i] - base) >> s2;
                atomicAdd(&row[bb >> 1], 1u << ((bb & 1u) << 4));
            }
        }
    }
    wave_lds_fence();
    KB_PH(7);
    // the number of entries below each bin, written back in place.  Lane l takes the 16-byte groups (8 bins) 32 j + l: the 32
    // lanes of a row read consecutive addresses (a lane that owned NB / 32 CONSECUTIVE bins, rounds 3-5, made every one of these
    // accesses a 16-way bank conflict at 2048 bins, and the LDS pipe is shared by the workgroup's 16 waves); the price is one
    // half-wave scan per j instead of one per row
    {
        constexpr int NJ = NB / 256;
        uint4 *grp = (uint4 *)row;
        uint4 t[NJ];
        unsigned carry = 0u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) t[j] = grp[32 * j + l5];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned gs = (t[j].x & 0xFFFFu) + (t[j].x >> 16) + (t[j].y & 0xFFFFu) + (t[j].y >> 16) + (t[j].z & 0xFFFFu) + (t[j].z >> 16) + (t[j].w & 0xFFFFu) + (t[j].w >> 16);
            const unsigned incl = kb_half_incl_scan_u32(gs);
            unsigned sacc = carry + incl - gs;
            const unsigned t0 = (unsigned)__builtin_amdgcn_readlane((int)incl, 31), t1 = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            carry += h ? t1 : t0;
            unsigned wd[4] = {t[j].x, t[j].y, t[j].z, t[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned lo = wd[e] & 0xFFFFu, hi = wd[e] >> 16;
                wd[e] = sacc | ((sacc + lo) << 16);
                sacc += lo + hi;
            }
            grp[32 * j + l5] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        }
    }
    wave_lds_fence();
    KB_PH(8);
    // lane j < k of each half: the bin that holds approximate rank j d = the last bin with at most j d entries below it; its
    // window is flagged, and the windows that neither overlap nor touch their predecessor open the runs
    unsigned F;
    {
        const bool win = l5 < (int)a.k;
        const unsigned m = (unsigned)l5 * a.d;
        unsigned lo = 0u, hi = (unsigned)(NB - 1);
        if (win) {
#pragma unroll 1
            for (int it = 0; it < LOGNB; ++it) {
                const unsigned mid = (lo + hi + 1u) >> 1;
                const bool le = (unsigned)cum16[mid] <= m;
                lo = le ? mid : lo;
                hi = le ? hi : mid - 1u;
            }
        }
        const unsigned L = lo > w ? lo - w : 0u, H = lo + w < (unsigned)(NB - 1) ? lo + w : (unsigned)(NB - 1);
        if (win) {
            for (unsigned wi = L >> 5; wi <= (H >> 5); ++wi) {
                const unsigned first = wi == (L >> 5) ? (L & 31u) : 0u, lastb = wi == (H >> 5) ? (H & 31u) : 31u;
                atomicOr(&bitmap[wi], (0xFFFFFFFFu >> (31u - lastb)) & (0xFFFFFFFFu << first));
            }
        }
        const unsigned prevH = dpp_get<0x111, 0xF>(H);               // row_shr:1 (lanes 0 and 32 read 0)
        const bool leader = win && (l5 == 0 || L > prevH + 1u);
        const unsigned long long lb = __ballot(leader);
        const unsigned lmask = h ? (unsigned)(lb >> 32) : (unsigned)lb;       // bit j: window j opens a run
        // the run's last window = the one before the next leader (or window k - 1)
        const unsigned above = l5 < 31 ? lmask >> (l5 + 1) : 0u;
        const unsigned nextl = above ? (unsigned)l5 + 1u + (unsigned)__builtin_ctz(above) : a.k;
        const unsigned Hend = (unsigned)__shfl((int)H, (int)(nextl - 1u) + 32 * h);
        unsigned cs = 0u, len = 0u;
        if (leader) {
            cs = cum16[L];
            const unsigned ce = Hend + 1u < (unsigned)NB ? (unsigned)cum16[Hend + 1u] : T;
            len = ce - cs;
        }
        const unsigned incl = kb_half_incl_scan_u32(len);
        const unsigned fs = incl - len;
        if (leader) rt[__popc(lmask & ((1u << l5) - 1u))] = make_uint2(L | (cs << 16), fs | (len << 16));
        F = (unsigned)__shfl((int)incl, 31 + 32 * h);                // the row's finalists = the entries of all its runs
    }
    const unsigned Fo = (unsigned)__shfl_xor((int)F, 32);
    const unsigned Fmax = F > Fo ? F : Fo;
    if (Fmax > (unsigned)KB_MAXFIN) return 0xFFFFFFFEu;
#if defined(KB_ABLATE) && KB_ABLATE == 2
    return F + Fo;
#endif
    wave_lds_fence();                                              // (the bins are dead from here: their words take the finalists)
    KB_PH(9);
    // collect: every entry of a flagged bin joins the finalist list (register-major order)
    {
        unsigned Fc = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (i < nv) {
                const unsigned bb = (v[i] - base) >> s2;
                const bool take = (unsigned)(32 * i) < tl5 && ((bitmap[(bb >> 5) & (unsigned)(KB_SCR_BITMAP - 1)] >> (bb & 31u)) & 1u);
                const unsigned long long bl = __ballot(take);
                if (bl != 0ull) {
                    if (take) fin[(Fc + kb_half_prefix(bl, h)) & (unsigned)(KB_MAXFIN - 1)] = v[i];
                    Fc += kb_half_count(bl, h);
                }
            }
        }
    }
    wave_lds_fence();
    if (l5 < 16) rcnt[l5] = 0u;                                     // (the bitmap is dead: its first words count the slots taken per run)
    wave_lds_fence();
    KB_PH(10);
#if defined(KB_ABLATE) && KB_ABLATE == 1
    return F + Fo;
#endif
    const float4 *q4 = (const float4 *)qbuf;
    bool bound_broken = false;
    for (unsigned f0 = 0; f0 < Fmax; f0 += 32) {
        const unsigned f = f0 + (unsigned)l5;
        const unsigned ent = f < F ? fin[f] : base;
        const unsigned idx = ent & 0xFFFu;
        const size_t vc = a.room_row0 + idx;
        const float sqj = a.sq[vc];
        // The finalist's 256-byte feature row, fetched by the QUAD (round 6): lane g of a quad loads bytes [64 s + 16 g, + 16) of
        // the row of quad member phi, for (phi, s) in 4 x 4 - one 64-byte sector per instruction and finalist - and a 4 x 4
        // transpose over (lane of the quad, register) then hands every lane its own finalist's row in ascending k.  One lane
        // per row (rounds 3-5: 64 different rows per wave instruction, sixteen 16-byte requests into each) cost 19 cycles of
        // the CU's vector-memory path per finalist, the quad form 9.4 with the transpose included, same bits
        // (tools/finalist_load_probe.hip: the path looks up one 64-byte piece per cycle whatever its lanes use of it; groups of
        // lanes 16 apart - gfx950's v_permlane swaps would transpose those in half the instructions - gain nothing).
        float4 C[16];
        {
            const int g = lane & 3;
            unsigned id4[4];
            id4[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)idx, 0x00, 0xF, 0xF, true);      // quad_perm [0,0,0,0]
            id4[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)idx, 0x55, 0xF, 0xF, true);
            id4[2] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)idx, 0xAA, 0xF, 0xF, true);
            id4[3] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)idx, 0xFF, 0xF, 0xF, true);
#pragma unroll
            for (int phi = 0; phi < 4; ++phi) {
                const float4 *pq = (const float4 *)(a.x + (a.room_row0 + id4[phi]) * a.ld) + g;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) C[4 * s4 + phi] = pq[4 * s4];
            }
        }
        // the finalist's run: the last one whose first bin is at most the finalist's (runs beyond the row's last read as
        // 0xFFFF), and the next free slot of the run's stretch of the composite array
        const unsigned bbf = (ent - base) >> s2;
        unsigned rho = 0u;
#pragma unroll
        for (int st = 8; st >= 1; st >>= 1) {
            const unsigned cand = rho + (unsigned)st;
            rho = (cand < 16u && (rt[cand & 15u].x & 0xFFFFu) <= bbf) ? cand : rho;
        }
        unsigned p = (unsigned)KB_MAXFIN;
        if (f < F) p = (rt[rho].y & 0xFFFFu) + atomicAdd(&rcnt[rho], 1u);
        {
            // in: lane g holds piece g of quad member phi's sector s4 in C[4 s4 + phi]; out: C[4 s4 + p] = piece p of the lane's own row
            const bool odd = lane & 1, up = lane & 2;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float *fc = (float *)&C[4 * s4];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float a0 = fc[c4], a1 = fc[4 + c4], a2 = fc[8 + c4], a3 = fc[12 + c4];
                    const float n0 = kb_quad_xor1(a0), n1 = kb_quad_xor1(a1), n2 = kb_quad_xor1(a2), n3 = kb_quad_xor1(a3);
                    const float b0 = odd ? n1 : a0, b1 = odd ? a1 : n0, b2 = odd ? n3 : a2, b3 = odd ? a3 : n2;
                    const float m0 = kb_quad_xor2(b0), m1 = kb_quad_xor2(b1), m2 = kb_quad_xor2(b2), m3 = kb_quad_xor2(b3);
                    fc[c4] = up ? m2 : b0; fc[4 + c4] = up ? m3 : b1; fc[8 + c4] = up ? b2 : m0; fc[12 + c4] = up ? b3 : m1;
                }
            }
        }
        // ascending-k fmaf chain from 0, the exact kernel's arithmetic (and a CPU sgemm's)
        float z = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 qv = q4[j];
            z = __fmaf_rn(qv.x, C[j].x, z); z = __fmaf_rn(qv.y, C[j].y, z); z = __fmaf_rn(qv.z, C[j].z, z); z = __fmaf_rn(qv.w, C[j].w, z);
        }
        // (|x_i|^2 + (-2 x_i.x_j)) + |x_j|^2, torch_edge.py:41-43, as in the exact kernel
        const float dd = __fadd_rn(__fmaf_rn(-2.0f, z, sq), sqj);
        if (p < (unsigned)KB_MAXFIN)
            sorted[p] = ((unsigned long long)rho << 60) | ((unsigned long long)key_of(dd) << 12) | (unsigned long long)idx;
        // The bound everything above rests on, L <= D <= L + 2 e_ij, assumes how the bf16 MFMA rounds internally (measured,
        // not documented).  Both sides are in registers here, so it is CHECKED for every finalist: the exact distance in key
        // units must lie in [key - 2, key + margin] (2 = the key's own rounding); a violation (or a NaN) sends the tile to the
        // exact path like any other failure, instead of ranking on a bound that did not hold.
        const float dk = __fmaf_rn(dd, inv_unit, KB_OFF), kf = (float)(ent >> 12);
        bound_broken |= f < F && !(dk >= kf - 2.0f && dk <= kf + (float)margin + 1.0f);
    }
    if (__ballot(bound_broken) != 0ull) return 0xFFFFFFFDu;
    wave_lds_fence();
    KB_PH(11);
#if defined(KB_ABLATE) && KB_ABLATE == 4
    return F + Fo;
#endif
    int32_t *out = a.out + (size_t)(h * KB_WAVES) * a.k;
    const unsigned long long m44 = (1ull << 44) - 1ull;
    for (unsigned f0 = 0; f0 < Fmax; f0 += 32) {
        const unsigned p = f0 + (unsigned)l5;
        const bool live = p < F;
        const unsigned long long x = live ? sorted[p] : 0ull;
        const unsigned long long xk = x & m44;
        const uint2 re = rt[(unsigned)(x >> 60)];
        const unsigned fs = re.y & 0xFFFFu, len = live ? re.y >> 16 : 0u;
        // exact rank = the run's first rank + the run's composites below x
        unsigned below = 0u;
        const unsigned span = wave_max_u32(len);
        for (unsigned j = 0; j < span; ++j) {
            const bool act = j < len;
            const unsigned long long y = sorted[act ? fs + j : p & 255u] & m44;
            below += (act && y < xk) ? 1u : 0u;
        }
        const unsigned rank = (re.x >> 16) + below;
        const unsigned t = (rank * a.magic) >> 18;
        if (live && t * a.d == rank && t < a.k) out[t] = (int32_t)(x & 0xFFFull);
    }
    KB_PH(12);
    return F + Fo;
}

// one candidate tile's nine operand fragments (k-steps 0..3 hi, 4..7 lo, 8 augmented), lane-major 16-byte pieces
__device__ __forceinline__ void kb_load_tile(kb_u32x4 (&fr)[9], const kb_u32x4 *__restrict__ p)
{
#pragma unroll
    for (int s = 0; s < 9; ++s) fr[s] = p[s * 64];
}
// acc (see the header) of the tile against the workgroup's 32 queries: 13 chained MFMAs
__device__ __forceinline__ kb_f32x16 kb_tile_product(const kb_u32x4 (&fr)[9], const kb_u32x4 (&bq)[8], const kb_u32x4 qa)
{
    kb_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the small products first (see the header: the accumulation error of an instruction scales with the partial sums it meets)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const kb_bf16x8 ch = __builtin_bit_cast(kb_bf16x8, fr[s]), cl = __builtin_bit_cast(kb_bf16x8, fr[4 + s]);
        const kb_bf16x8 qh = __builtin_bit_cast(kb_bf16x8, bq[s]), ql = __builtin_bit_cast(kb_bf16x8, bq[4 + s]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, ql, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl, qh, acc, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kb_bf16x8, fr[s]), __builtin_bit_cast(kb_bf16x8, bq[s]), acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kb_bf16x8, fr[8]), __builtin_bit_cast(kb_bf16x8, qa), acc, 0, 0, 0);
}

#ifndef KB_MIN_WAVES_EU
#define KB_MIN_WAVES_EU 4
#endif
__global__ __launch_bounds__(KB_WAVES * 64, KB_MIN_WAVES_EU) void knn_bf_kernel(KnnBfArgs a)
{
    constexpr int CAP = KB_CAP, ROW = KB_ROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    unsigned *ent = (unsigned *)kb_smem;                      // [KB_Q][ROW]
    float *thr_f = (float *)(ent + KB_Q * ROW);               // [KB_Q] admission threshold on acc
    unsigned *cnt = (unsigned *)(thr_f + KB_Q);               // [KB_Q]
    unsigned *tau = cnt + KB_Q;                               // [KB_Q] key threshold of the row
    unsigned *keyx = tau + KB_Q;                              // [KB_Q] keyX of the cut: KK entries at or below it prove the row complete
    float *rpar = (float *)(keyx + KB_Q);                     // [KB_Q][KB_RP] row parameters
    float *scr = rpar + KB_Q * KB_RP + 2 * KB_SCR * (threadIdx.x >> 6);   // [KB_WAVES][2][KB_SCR] per wave and row: query features, bitmap, run table
    unsigned *smax = (unsigned *)(rpar + KB_Q * KB_RP + 2 * KB_SCR * KB_WAVES);   // [KB_WAVES]
    unsigned *fail = smax + KB_WAVES;                         // [4] (one used)
    unsigned *sink = fail + 4 + threadIdx.x;                  // [KB_WAVES * 64] one word per thread: entries that fail the test

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order: blocks b, b + 8, .. share an XCD (round-robin dispatch), so XCD x takes the contiguous tiles
    // [x G / 8, (x + 1) G / 8): a room's operands are fetched into one or two L2s instead of all eight (speed only)
    const unsigned G = gridDim.x, b = blockIdx.x;
    const int tile = (int)((G & 7u) == 0u ? (b & 7u) * (G >> 3) + (b >> 3) : b);
    const int tpr = a.N >> 5;                                 // 32-vertex tiles per room
    const int room = tile / tpr;
    const size_t room_row0 = (size_t)room * a.N;
    const size_t tile0 = (size_t)room * tpr;
    const unsigned KK = (unsigned)a.KK;
    // wave w streams the tiles w n_steps .. w n_steps + n_steps - 1: step 0 of the workgroup is a strided sample of the room
    const int n_steps = (tpr + KB_WAVES - 1) / KB_WAVES;
    const int nv0 = (tpr + n_steps - 1) / n_steps;            // waves that hold a tile at step 0 = sample registers per lane of the cut

    {
        unsigned mb = 0u;
        const float4 *sq4 = (const float4 *)(a.sq + room_row0);
        for (int i = tid; i < (a.N >> 2); i += KB_WAVES * 64) {
            const float4 s4 = sq4[i];
            const unsigned b0 = __float_as_uint(s4.x) & 0x7FFFFFFFu, b1 = __float_as_uint(s4.y) & 0x7FFFFFFFu;
            const unsigned b2 = __float_as_uint(s4.z) & 0x7FFFFFFFu, b3 = __float_as_uint(s4.w) & 0x7FFFFFFFu;
            const unsigned m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3, m = m01 > m23 ? m01 : m23;
            mb = m > mb ? m : mb;
        }
        mb = wave_max_u32(mb);
        if (lane == 0) smax[wave] = mb;
    }
    if (tid == 0) *fail = 0u;
    // the tile's queries as the B operand: hi (k-steps 0..3) and lo (4..7) fragments
    kb_u32x4 bq[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) bq[s] = a.bp[((size_t)tile * 9 + s) * 64 + lane];
    // the wave's sample tile is requested here, so that its latency runs under the prologue
    kb_u32x4 fr0[9];
    if (wave < nv0) kb_load_tile(fr0, a.bp + (tile0 + (size_t)(wave * n_steps)) * 9 * 64 + lane);
#ifdef PSG_KF_TL
    unsigned long long tl0 = 0;
    if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl0) :: "memory");
#endif
    __syncthreads();
    const unsigned mbits = wave_max_u32(smax[lane & (KB_WAVES - 1)]);
    // (bit patterns: NaN / inf / huge and zero / tiny rooms take the exact path)
    bool need_exact = !(mbits >= 0x20000000u && mbits <= 0x7C000000u);
    const float sqmaxf = __uint_as_float(mbits);
    if (tid < KB_Q) {
        // row parameters: the key unit is E_i(|x_i|^2) / KB_UNIT_DIV (a query far below the room's norms gets a floor, so that the
        // unit stays a normal number)
        const float sqi = a.sq[(size_t)tile * KB_Q + tid];
        const float sqe = fmaxf(sqi, sqmaxf * (1.0f / 4096.0f));
        const float ae = sqrtf(sqe), t0 = 2.0298f * ae;
        const float unit = (2.0f / (float)KB_UNIT_DIV) * (KB_A * ae * t0 + KB_B * t0 * t0 + KB_G * sqe);
        const float inv_unit = 1.0f / unit;
        float *rp = rpar + tid * KB_RP;
        rp[0] = sqi; rp[1] = sqrtf(sqi); rp[2] = unit; rp[3] = inv_unit;
        rp[4] = __fmaf_rn(sqi * (1.0f - KB_G), inv_unit, KB_OFF);      // key = round(acc * (-2 / u) + this)
        cnt[tid] = (unsigned)(32 * nv0 * a.nsamp);
        tau[tid] = KB_TAUMAX;
        keyx[tid] = 0u;
        // step 0 takes everything inside the key range (acc * m2 + ci < KEYMAX - 1, two units for the evaluation)
        thr_f[tid] = (rp[4] - (float)(KB_KEYMAX - 2u)) * (0.5f * unit);
    }
    __syncthreads();
    KB_TL(0);
    const float m2 = -2.0f * rpar[r * KB_RP + 3];
    const float ci23 = rpar[r * KB_RP + 4] + 8388608.0f;      // key = mantissa of acc * m2 + ci23
    unsigned long long st_fin = 0, st_ent = 0;
    KbRowArgs ra;
    ra.row_dw = (unsigned)(wave * ROW); ra.rp_dw = (unsigned)(rpar - (float *)kb_smem);
    ra.cnt_dw = (unsigned)(cnt - (unsigned *)kb_smem); ra.thr_dw = (unsigned)(thr_f - (float *)kb_smem);
    ra.tau_dw = (unsigned)(tau - (unsigned *)kb_smem); ra.keyx_dw = (unsigned)(keyx - (unsigned *)kb_smem);
    ra.KK = KK; ra.wave = wave;
    ra.T[0] = ra.T[1] = 0u;

    if (!need_exact) {
        // the augmented k-step's query side: 1, 1, 1 (x the three pieces of -(1 - B) |x_j|^2 / 2) and |x_i| rounded up to bf16
        // (x (A / 2) |x_j| rounded up) in elements 0..3 of the h = 0 lanes
        kb_u32x4 qa = kb_u32x4{0u, 0u, 0u, 0u};
        if (!h) {
            const unsigned au = __float_as_uint(rpar[r * KB_RP + 1]);
            qa = kb_u32x4{0x3F803F80u, 0x00003F80u | (((au + 0xFFFFu) >> 16) << 16), 0u, 0u};
        }
        // the sample: every pair's entry at a fixed slot (sample step, wave, half, g); pairs beyond the key range (and NaNs) get
        // the key KEYMAX, which no threshold keeps.  Long rows (nsamp = 2: KK > 310, d >= 21) sample two tiles per wave, the
        // wave's first and the one half way through its range: the threshold's rank is then known twice as well relative
        // to its size, rows end with ~1.45 KK entries instead of 1.7 KK and stay inside their 1024-entry buffers
        const int nsamp = a.nsamp, t_half = n_steps >> 1;
        auto store_sample = [&](const kb_f32x16 acc, int t, int ss) {
            const float th = thr_f[r];
            const unsigned cb = (unsigned)(32 * t + 4 * h);
            unsigned *dst = ent + r * ROW + (ss * KB_WAVES + wave) * 32 + h * 16;
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
                unsigned e[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned kb = __float_as_uint(__fmaf_rn(acc[g + u], m2, ci23));
                    const unsigned key = acc[g + u] >= th ? (kb & 0xFFFFFu) : KB_KEYMAX;
                    e[u] = (key << 12) | cb | (unsigned)(((g + u) & 3) + 8 * ((g + u) >> 2));
                }
                *(uint2 *)(dst + g) = make_uint2(e[0], e[1]);
            }
        };
        if (wave < nv0) {
            const int t = wave * n_steps;
            const kb_f32x16 acc = kb_tile_product(fr0, bq, qa);
            if (nsamp == 2) kb_load_tile(fr0, a.bp + (tile0 + t + t_half) * 9 * 64 + lane);
            store_sample(acc, t, 0);
            if (nsamp == 2) store_sample(kb_tile_product(fr0, bq, qa), t + t_half, 1);
        }
        __syncthreads();
        KB_TL(1);
        {
            const bool bad = nsamp == 2 ? kb_cut_sample<32>(ra, lane, 2 * nv0, a.rsel, a.fine_cut != 0)
                                        : kb_cut_sample<16>(ra, lane, nv0, a.rsel, a.fine_cut != 0);
            if (__ballot(bad) != 0ull && lane == 0) atomicOr(fail, 1u);
        }
        __syncthreads();
        KB_TL(2);
        need_exact = *fail != 0u;
        if (!need_exact && n_steps > 1) {
            // the stream: lane (query r, half h) holds the 16 candidates 32 t + (g & 3) + 8 (g >> 2) + 4 h of tile t.  The next
            // tile's nine operand fragments are requested as soon as the products have consumed this tile's, so that their
            // latency runs under the append work (requested one by one in front of the MFMAs that use them, each wave waited
            // nine L2 round trips per tile)
            const float th = thr_f[r];
            const int t0w = wave * n_steps;
            const int n_w = t0w + n_steps <= tpr ? n_steps : (tpr > t0w ? tpr - t0w : 0);       // tiles of this wave
            const int skip = nsamp == 2 ? t_half : -1;                                          // (its second sample tile)
            int j = 1 == skip ? 2 : 1;
            kb_u32x4 fr[9];
            if (j < n_w) kb_load_tile(fr, a.bp + (tile0 + t0w + j) * 9 * 64 + lane);
            while (j < n_w) {
                const int t = t0w + j;
                const kb_f32x16 acc = kb_tile_product(fr, bq, qa);
                j = j + 1 == skip ? j + 2 : j + 1;
                if (j < n_w) kb_load_tile(fr, a.bp + (tile0 + t0w + j) * 9 * 64 + lane);
                const unsigned cb = (unsigned)(32 * t + 4 * h);
                unsigned n = 0;
#pragma unroll
                for (int g = 0; g < 16; ++g) n += acc[g] >= th ? 1u : 0u;
                if (n) {
                    const unsigned base = atomicAdd(&cnt[r], n);
                    if (base + n <= (unsigned)CAP) {
                        // branch-free append: every pair builds its entry, the ones that fail the test store it to the
                        // thread's sink word (16 exec-masked blocks with two taken branches each cost more)
                        unsigned *dst = ent + r * ROW + base;
                        unsigned off = 0;
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            const bool pass = acc[g] >= th;
                            // key = round(acc * m2 + ci) read off the mantissa of (.. + 2^23); << 12 drops the exponent bits
                            const unsigned kb = __float_as_uint(__fmaf_rn(acc[g], m2, ci23));
                            unsigned *p = pass ? dst + off : sink;
                            *p = (kb << 12) | cb | (unsigned)((g & 3) + 8 * (g >> 2));
                            off += pass ? 1u : 0u;
                        }
                    }
                }
            }
        }
        __syncthreads();
        KB_TL(8);
        // an overflowing row, or one that the key range left with fewer than KK candidates: exact path
        unsigned Twg = 0u;
        if (!need_exact) {
            const unsigned cl = cnt[lane & (KB_Q - 1)];
            const bool over = __ballot(cl > (unsigned)CAP) != 0ull, under = __ballot(cl < KK) != 0ull;
            need_exact = over || under;
            if (need_exact && tid == 0) atomicOr(fail, (over ? 2u : 0u) | (under ? 4u : 0u));
            Twg = wave_max_u32(cl);
        }
        if (!need_exact) {
            KbFinalArgs fa;
            fa.x = a.x; fa.ld = (unsigned)a.ld; fa.sq = a.sq; fa.k = (unsigned)a.k; fa.d = (unsigned)a.d; fa.magic = a.magic;
            fa.room_row0 = room_row0; fa.qglob = (size_t)tile * KB_Q + wave;
            fa.out = a.out + fa.qglob * a.k;
            fa.scr_dw = (unsigned)(scr - (float *)kb_smem);
            ra.T[0] = cnt[wave]; ra.T[1] = cnt[wave + KB_WAVES];
            const unsigned Tm = ra.T[0] > ra.T[1] ? ra.T[0] : ra.T[1];
            const int nv = (int)((Tm + 31u) >> 5);
            // (the bin count follows the workgroup's longest row, so every wave runs the same code at the same time)
            const unsigned nf = Twg <= 256u ? kb_final_rows<8, 256>(ra, fa, lane, nv)
                              : (Twg <= 512u ? kb_final_rows<16, KB_LONG_BINS>(ra, fa, lane, nv) : kb_final_rows<32, KB_LONG_BINS>(ra, fa, lane, nv));
            if (nf >= 0xFFFFFFFDu) { if (lane == 0) atomicOr(fail, 8u << (0xFFFFFFFFu - nf)); }
            else { st_fin += nf; st_ent += ra.T[0] + ra.T[1]; }
            KB_TL(9);
            __syncthreads();
            KB_TL(10);
#ifdef PSG_KF_TL
            if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); atomicMax(&g_kf_tl[11], t_ - tl0); g_kf_tl[27] = 1ull; }
#endif
            need_exact = *fail != 0u;
        }
    }
    if (a.stats) {
        // (diagnostic) per-workgroup sums through LDS, then one set of global atomics per workgroup
        __syncthreads();
        unsigned *acc3 = (unsigned *)smax;
        if (tid < 3) acc3[tid] = 0u;
        __syncthreads();
        if (lane == 0 && !need_exact) { atomicAdd(&acc3[0], (unsigned)st_fin); atomicAdd(&acc3[1], 2u); atomicAdd(&acc3[2], (unsigned)st_ent); }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&a.stats[0], 1ull);
            if (need_exact) {
                atomicAdd(&a.stats[1], 1ull);
                // why (16-bit fields): [6] cut without a threshold | a row above CAP | a row short of KK entries; [7] final: a row short of
                // KK keys below keyX | more than 256 finalists | a finalist outside the proven bound
                const unsigned fb = *fail;
                atomicAdd(&a.stats[6], (unsigned long long)(fb & 1u) | ((unsigned long long)((fb >> 1) & 1u) << 16) | ((unsigned long long)((fb >> 2) & 1u) << 32));
                atomicAdd(&a.stats[7], (unsigned long long)((fb >> 3) & 1u) | ((unsigned long long)((fb >> 4) & 1u) << 16) | ((unsigned long long)((fb >> 5) & 1u) << 32));
            }
            else {
                atomicAdd(&a.stats[2], (unsigned long long)KB_Q); atomicAdd(&a.stats[3], (unsigned long long)acc3[0]);
                atomicAdd(&a.stats[4], (unsigned long long)acc3[1]); atomicAdd(&a.stats[5], (unsigned long long)acc3[2]);
            }
        }
    }
    if (need_exact) {
        // the exact path for the tile's two 16-query blocks (results identical where the fast path had already written)
        __syncthreads();
        knn_exact_block_cold(a.exact, 2 * tile);
        __syncthreads();
        knn_exact_block_cold(a.exact, 2 * tile + 1);
    }
}

inline size_t knn_bf_lds_bytes()
{
    const size_t own = (size_t)KB_Q * KB_ROW * 4 + KB_Q * 16 + KB_Q * KB_RP * 4 + KB_WAVES * 2 * KB_SCR * 4 + KB_WAVES * 4 + 16 + KB_WAVES * 64 * 4;
    return own > knn_fused_lds_bytes() ? own : knn_fused_lds_bytes();
}
