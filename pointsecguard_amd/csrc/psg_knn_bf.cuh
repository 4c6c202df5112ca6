// Dilated kNN graph in feature space (C = 64) with a bf16 PREFILTER on the matrix cores: the exact fp32 distance - the one
// that decides the reference's neighbour order, ResGCN/gcn_lib/dense/torch_edge.py:32-59 - is evaluated only for the few
// candidates per row that can hold one of the wanted ranks; everything else is decided on a LOWER BOUND of the distance
// with a proven, per-pair error term.  Same results, bit for bit, as the exact kernel (psg_knn_fused.cuh), which stays in
// the same launch as the fallback for every tile whose bound cannot be kept.
//
// Approximation.  x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|.  One 32 x 32 tile of
//     acc = (hi_i.hi_j + hi_i.lo_j + lo_i.hi_j) - (1 - B) |x_j|^2 / 2 + (A / 2) |x_i||x_j|
// is 13 v_mfma_f32_32x32x16_bf16 (K = 3 x 64 + one augmented step): 3/16 of the fp32 MFMA time of the exact kernel.
// L = (1 - G) |x_i|^2 - 2 acc = D~ - e_ij with D~ the approximate distance and e_ij = A |x_i||x_j| + B |x_j|^2 + G |x_i|^2.
// Error of D~ against the reference's fp32 distance D (psg_knn_ops.cuh builds the operands): dropped products (lo.lo and
// the split residues) <= 3 * 2^-18 |x_i||x_j|; fp32 accumulation of 13 instructions of 16 exact products each, every
// addition rounded to nearest: <= 13 * 17 * 2^-24 (|x_i||x_j| + |x_j|^2 / 2); the reference's own chain 64 * 2^-24
// |x_i||x_j| and its two distance roundings <= 2^-23 (|x_i|^2 + |x_j|^2) + 2^-22 |x_i||x_j|: in distance units (dot errors
// count twice) |D~ - D| <= 2^-14.1 |x_i||x_j| + 2^-16.2 |x_j|^2 + 2^-23 |x_i|^2.  The kernel uses A = 2^-13.5, B = 2^-15.5,
// G = 2^-20 (1.5 x that bound; tools/bf16_split_probe.hip measured 2^-16 of the largest |x|^2 on hardware over eight data
// sets: random, offset, 1e6 dynamic range, near-cancelling pairs, 1e-15 and 1e15 magnitudes), so L <= D <= L + 2 e_ij.
// The error scales with the PAIR's norms, not with the room's largest norm: the network's own features have a few points
// whose |x|^2 is 40 x the mean, and a bound in units of that maximum was wider than the neighbour distances themselves.
//
// Row margin.  For every candidate j with L_ij <= T (a threshold in distance units): (|x_j| - |x_i|)^2 <= D <= T + 2 e_ij,
// hence |x_j| <= 1.0148 |x_i| + 1.015 sqrt(T) and 2 e_ij <= E_i(T) = 2 (A |x_i| t + B t^2 + G |x_i|^2) at that t: one
// number per row and threshold bounds the gap between L and D for everything the row can hold.
//
// Keys.  key = round((L + 16 u_i) / u_i) with the row's own unit u_i = E_i(|x_i|^2) / 8 ~ 2^-13.9 |x_i|^2 (one FMA whose
// constant carries 2^23: the key is the result's mantissa); a row entry is key << 12 | index (32 bits: twice the rows of
// the exact kernel in the same LDS).  Candidates farther than 2^20 units (71 |x_i|^2) are never admitted; a row that
// needs them ends with fewer than KK entries and takes the exact path.
//
// Stream (as in the exact kernel: 16 waves, shared row buffers, slots by one LDS atomic per lane and step, cuts on a
// data-independent schedule, no barrier in between) for 32 queries per workgroup; a candidate is admitted by one float
// compare acc >= th_i (a superset of key <= tau_i).  A cut finds thr with #(keys <= thr) >= KK (the histogram bin in
// which the cumulative count reaches KK); at least KK candidates then have D <= thr u + E_i(thr u), so the true KK-th
// distance is below that and every candidate that matters has L below it: tau = thr + E_i / u keeps them all.  A counter
// above CAP (adversarial orders, massive ties) sends the tile to the exact path.
//
// Final ranks.  Wanted: exact ranks 0, d, .., (k-1) d.  With l_m the m-th smallest L of the row, l_m <= D_(m) <= l_m + E,
// so the element of exact rank m has its L within E of l_m; everything with L below l_m - E is smaller than it,
// everything above l_m + E larger.  Entries are binned by key (bin 0 = the query's own neighbourhood, bins 1.. of a
// power-of-two width: exact integer edges), the bin that holds approximate rank m and w = ceil(E / width) bins to each side
// are flagged, overlapping windows merge into RUNS, the members of flagged bins ("finalists", a few per wanted rank) get
// their exact distance (ascending-k fmaf chain from the fp32 copy, the exact kernel's arithmetic) and rank themselves
// inside their run: first rank of the run + the number of smaller finalists of the same run IS the exact rank.  More than
// 256 finalists in a row send the tile to the exact path.
//
// Cuts and final ranking run one row per HALF wave (the wave's rows wave and wave + 16 side by side, every wave of the
// workgroup at the same time): per-row reductions are 16-lane DPP steps plus one cross-row exchange and serve both rows
// with one instruction stream; what is left is per-entry work, 32 lanes per row.  (These phases are bound by vector
// instruction issue - 4 cycles per wave instruction and SIMD -, not by latency: a version with one row per wave, and
// one with the two rows interleaved, cost twice the instructions.)
#pragma once

constexpr int KB_WAVES = 16;
constexpr int KB_Q = 32;                      // query rows per workgroup (one 32-column MFMA tile)
constexpr int KB_CAP = 1024;                  // entries per row buffer
constexpr int KB_ROW = KB_CAP + 2;            // dwords between rows: 8-byte aligned, skewed over the LDS banks
constexpr float KB_A = KNN_BF_A, KB_B = KNN_BF_B, KB_G = KNN_BF_G;   // 2^-13.5, 2^-15.5, 2^-20 (psg_knn_ops.cuh)
constexpr float KB_OFF = 16.0f;               // key offset in units: L of the query itself (>= -2 e_ii ~ -3.2 u) stays positive
constexpr unsigned KB_KEYMAX = 0xFFFFEu;      // largest admissible key
constexpr int KB_MAXFIN = 256;
constexpr int KB_RP = 8;                      // floats of row parameters: |x|^2, |x|, unit, 1 / unit, key constant, spare x3

typedef __bf16 kb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float kb_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned kb_u32x4 __attribute__((ext_vector_type(4)));     // one 16-byte operand fragment

// Diagnostic build (EXTRA=-DPSG_KF_TL, tools/knn_timeline.py): thread 0 adds the time since kernel start at the workgroup's
// barrier exits into the exact kernel's g_kf_tl table (same slots: 0 start, 1-6 first three cuts, 8 stream end, 9 wave 0
// ranked, 10 all ranked)
#ifdef PSG_KF_TL
#define KB_TL(idx) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    atomicAdd(&g_kf_tl[(idx)], t_ - tl0); atomicAdd(&g_kf_tl[16 + (idx)], 1ull); } } while (0)
__device__ unsigned long long g_kb_ph[32];      // phase cycle sums of the final ranking / cuts (wave 0 of every workgroup)
#define KB_PH_BEGIN unsigned long long ph_t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_t0_) :: "memory")
#define KB_PH(idx) do { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (threadIdx.x == 0) { atomicAdd(&g_kb_ph[(idx)], t_ - ph_t0_); atomicAdd(&g_kb_ph[16 + (idx)], 1ull); } ph_t0_ = t_; } while (0)
#else
#define KB_TL(idx)
#define KB_PH_BEGIN
#define KB_PH(idx)
#endif

struct KnnBfArgs {
    const kb_u32x4 *bp;   // [rows / 32][9][64] fragments (psg_knn_ops.cuh)
    const float *sq;      // [rows]
    int32_t *out;         // [rows][k]
    int N;                // points per room, multiple of 32
    int k, d, KK;
    unsigned magic;       // ceil(2^18 / d)
    int LOW;              // a cut shortens the rows that hold more than LOW entries
    int first_cut;        // candidates seen at the first cut (a multiple of 512)
    float grow;           // a cut at n candidates seen is followed by one at grow * n
    KnnFusedArgs exact;            // the exact path's arguments (xp, sq, out, ...)
    unsigned long long *stats;     // optional [8]: tiles, fallback tiles, rows, finalists, cut events, entries at the end
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

// E_i(T) / u_i: the gap between L and D, in key units, for every candidate of the row with L <= T (see the header)
__device__ __forceinline__ unsigned kb_margin_keys(float a, float sq, float inv_unit, float T)
{
    const float t = 1.0148f * a + 1.015f * sqrtf(fmaxf(T, 0.0f));
    const float E = 2.0f * (KB_A * a * t + KB_B * t * t + KB_G * sq);
    const float m = E * inv_unit * 1.0001f + 2.0f;        // + quantisation of the two keys compared
    return m < 1.0e6f ? (unsigned)m + 1u : 1000000u;
}

// ---- cuts and final ranking: the wave's two rows side by side, one per half wave; lane l of a half holds the entries
// 32 i + l of its row (slots beyond T hold the sentinel 0xFFFFFFFF: real entries are below 2^31); the row's own LDS is
// scratch once the entries are in registers.  (Not inlined: with them and the exact path inlined the kernel spilled inside
// its stream loop.  LDS locations travel as dword offsets into the workgroup's dynamic LDS, so that every access stays a ds_
// instruction: a pointer parameter would be a generic pointer and turn them into flat_ accesses.)
struct KbRowArgs {
    unsigned row_dw;        // the first row's entries; the second row's are 16 rows further
    unsigned rp_dw;         // row parameters [KB_Q][KB_RP] floats
    unsigned cnt_dw, thr_dw, tau_dw;   // per-row counters, admission thresholds on acc, key thresholds
    unsigned T[2];          // entries held (0: the row takes no part)
    unsigned KK;
    int wave;
};

// Cut: 256 linear bins of 32-bit counters over the entry values above the query's own neighbourhood (bin 0); the bin in
// which the cumulative count reaches KK gives thr; tau = thr + E_i(thr) / u_i (+ 2 for the float bin edges); every entry
// with key <= tau is kept, compacted in place.  Lane 0 of each half stores the row's new count and thresholds.
template <int NV>
__device__ __noinline__ void kb_cut_rows(const KbRowArgs c, int lane)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    const int h = lane >> 5, l5 = lane & 31;
    const int ridx = c.wave + KB_WAVES * h;
    unsigned *row = (unsigned *)kb_smem + c.row_dw + h * KB_WAVES * KB_ROW;
    const float *rp = (const float *)kb_smem + c.rp_dw + ridx * KB_RP;
    const unsigned T = h ? c.T[1] : c.T[0];
    unsigned v[NV];
    unsigned mx1 = 0u;
    KB_PH_BEGIN;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned e = (unsigned)(i * 32 + l5);
        const unsigned x = row[e];
        v[i] = e < T ? x : 0xFFFFFFFFu;
        mx1 = v[i] + 1u > mx1 ? v[i] + 1u : mx1;               // (the sentinel wraps to 0)
    }
    const float sq = rp[0], a = rp[1], unit = rp[2], inv_unit = rp[3], ci = rp[4];
    wave_lds_fence();
    KB_PH(0);
    const unsigned mx = kb_half_max_u32(mx1) - 1u;
    // the query's own neighbourhood: keys up to offset + E_i(0) / u_i
    const unsigned lowE = (((unsigned)KB_OFF + kb_margin_keys(a, sq, inv_unit, 0.0f)) << 12) | 0xFFFu;
    const unsigned mn2 = mx > lowE ? lowE : mx;
    const float span = (float)(mx - mn2 + 1u);
    const float scale = 254.0f / span, inv_scale = span * (1.0f / 254.0f);
    // 256 counters per row; the lane owns bins 8 l .. 8 l + 7.  (A half whose row takes no part, T = 0, must leave that row's
    // entries alone: nothing of it is in registers.)
    if (T != 0u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ((uint2 *)row)[4 * l5 + j] = make_uint2(0u, 0u);
    }
    wave_lds_fence();
    KB_PH(1);
    if (T != 0u) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned bb = 1u + (unsigned)((float)(v[i] - mn2) * scale);      // (sentinels land in bin 255 and are taken out below)
            bb = v[i] <= lowE ? 0u : (bb > 255u ? 255u : bb);
            atomicAdd(&row[bb], 1u);
        }
    }
    wave_lds_fence();
    KB_PH(2);
    unsigned cw[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint2 t = ((const uint2 *)row)[4 * l5 + j]; cw[2 * j] = t.x; cw[2 * j + 1] = t.y; }
    if (l5 == 31) cw[7] -= (unsigned)(NV * 32) - T;
    unsigned tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += cw[j];
    const unsigned incl = kb_half_incl_scan_u32(tot), start = incl - tot;
    unsigned mybin = (unsigned)(8 * l5), s = start;
#pragma unroll
    for (int j = 0; j < 7; ++j) { s += cw[j]; mybin += s < c.KK ? 1u : 0u; }
    const unsigned long long own = __ballot(T != 0u && start < c.KK && c.KK <= incl);       // one lane per half (KK <= T)
    const unsigned ownh = h ? (unsigned)(own >> 32) : (unsigned)own;
    const int owner = (ownh ? __builtin_ctz(ownh) : 0) + 32 * h;
    const unsigned binB = (unsigned)__shfl((int)mybin, owner);
    // an entry of a bin <= binB (>= 1) has 1 + (v - mn2) * scale < binB + 1 in float: v - mn2 < binB / scale * (1 + 2^-22), less
    // than one key above; key(thr) + 2 covers that and the floor of the shift
    const unsigned X = binB ? mn2 + (unsigned)((float)binB * inv_scale) : lowE;
    const unsigned keyX = (X >> 12) + 2u;
    const unsigned margin = kb_margin_keys(a, sq, inv_unit, ((float)keyX - KB_OFF) * unit);
    unsigned tauk = keyX + margin;
    tauk = tauk > KB_KEYMAX ? KB_KEYMAX : tauk;
    const unsigned keepmax = (tauk << 12) | 0xFFFu;
    KB_PH(3);
    unsigned base = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool keep = v[i] <= keepmax;                      // (sentinels are above any keepmax)
        const unsigned long long bl = __ballot(keep);
        if (keep) row[base + kb_half_prefix(bl, h)] = v[i];
        base += kb_half_count(bl, h);
    }
    if (l5 == 0 && T != 0u) {
        ((unsigned *)kb_smem)[c.cnt_dw + ridx] = base;
        ((unsigned *)kb_smem)[c.tau_dw + ridx] = tauk;
        // admit iff key <= tauk, i.e. acc * m2 + ci < tauk + 1; two more units cover the float evaluation
        ((float *)kb_smem)[c.thr_dw + ridx] = (ci - (float)(tauk + 3u)) * (0.5f * unit);
    }
    KB_PH(4);
}

// Final ranking (see the header).  Row LDS: 1024 bins of 16 bits (counts, then first ranks) + 256 finalists; scr = 96 words
// of wave-private LDS per row (the query's 64 features, 32 words of flagged-bin bitmap).  Returns the number of finalists of
// the two rows, or 0xFFFFFFFF when a row must take the exact path.
struct KbFinalArgs {          // by value: a reference to the kernel's argument struct would force it (and every pointer in it) through the stack
    const float *xp, *sq;
    int32_t *out;             // the first row's output; the second row's is 16 k further
    size_t room_row0, qglob;  // global row of the room's first point and of the first row's query (second: + 16)
    unsigned scr_dw, k, d, magic;
};

template <int NV>
__device__ __noinline__ unsigned kb_final_rows(const KbRowArgs c, const KbFinalArgs a, int lane)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    const int h = lane >> 5, l5 = lane & 31;
    const int ridx = c.wave + KB_WAVES * h;
    unsigned *row = (unsigned *)kb_smem + c.row_dw + h * KB_WAVES * KB_ROW;
    const float *rp = (const float *)kb_smem + c.rp_dw + ridx * KB_RP;
    float *qbuf = (float *)kb_smem + a.scr_dw + h * 96;
    unsigned *bitmap = (unsigned *)kb_smem + a.scr_dw + h * 96 + 64;
    const unsigned T = h ? c.T[1] : c.T[0];
    unsigned v[NV], bn[NV];
    unsigned mx1 = 0u;
    KB_PH_BEGIN;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned e = (unsigned)(i * 32 + l5);
        const unsigned x = row[e];
        v[i] = e < T ? x : 0xFFFFFFFFu;
        mx1 = v[i] + 1u > mx1 ? v[i] + 1u : mx1;
    }
    const float sq = rp[0], aa = rp[1], unit = rp[2], inv_unit = rp[3];
    const unsigned tau_row = ((const unsigned *)kb_smem)[c.tau_dw + ridx];
    // the query's 64 features in k order for the exact chains (feature k = 16 i + 4 e + g: element e of float4 [i][g])
    const size_t qg = a.qglob + (size_t)(h * KB_WAVES);
    {
        const float *xq = a.xp + ((qg >> 4) * 256 + (qg & 15)) * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kf = l5 + 32 * u;
            qbuf[kf] = xq[((kf >> 4) * 64 + 16 * (kf & 3)) * 4 + ((kf >> 2) & 3)];
        }
    }
    bitmap[l5] = 0u;
    wave_lds_fence();
    KB_PH(5);
    const unsigned kmx = (kb_half_max_u32(mx1) - 1u) >> 12;
    // the gap between L and D for everything the row holds: from its threshold, or (never cut) from its largest key
    const unsigned ktop = tau_row < kmx ? tau_row : kmx;
    const unsigned margin = kb_margin_keys(aa, sq, inv_unit, ((float)ktop - KB_OFF) * unit);
    const unsigned lowmax = (unsigned)KB_OFF + kb_margin_keys(aa, sq, inv_unit, 0.0f);      // the query's own neighbourhood: bin 0
    unsigned k2 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned kq = v[i] >> 12;
        k2 = (kq > lowmax && kq < k2) ? kq : k2;                    // (a sentinel's key 0xFFFFF is above every real key)
    }
    k2 = ~kb_half_max_u32(~k2);
    const unsigned range = (k2 <= kmx) ? kmx - k2 : 0u;
    unsigned sh = range > 1022u ? (unsigned)(32 - __builtin_clz(range)) - 10u : 0u;
    if ((range >> sh) > 1022u) ++sh;                                // bins 1 .. 1 + (range >> sh) <= 1023
    const unsigned w = (margin + (1u << sh) - 1u) >> sh;            // window half width in bins
    const unsigned c0 = (1u << sh) - k2;                            // bin = ((v >> 12) + c0) >> sh above lowmax
    const unsigned low_ent = (lowmax << 12) | 0xFFFu;
    unsigned short *cum16 = (unsigned short *)row;
    unsigned long long *fin = (unsigned long long *)(row + 512);
#pragma unroll
    for (int j = 0; j < 8; ++j) ((uint2 *)row)[8 * l5 + j] = make_uint2(0u, 0u);
    wave_lds_fence();
    KB_PH(6);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        unsigned bb = ((v[i] >> 12) + c0) >> sh;                  // (keys reach 2^20: no room to shift the entry itself)
        bb = v[i] <= low_ent ? 0u : (bb > 1023u ? 1023u : bb);   // (sentinels: bin 1023, taken out below)
        bn[i] = bb;
        atomicAdd(&row[bb >> 1], 1u + (bb & 1u) * 0xFFFFu);
    }
    wave_lds_fence();
    KB_PH(7);
    // the lane's own 32 bins (32 l .. 32 l + 31) -> the number of entries below each bin, written back in place
    {
        unsigned wd[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint2 t = ((const uint2 *)row)[8 * l5 + j]; wd[2 * j] = t.x; wd[2 * j + 1] = t.y; }
        if (l5 == 31) wd[15] -= ((unsigned)(NV * 32) - T) << 16;
        unsigned tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += (wd[j] & 0xFFFFu) + (wd[j] >> 16);
        unsigned s = kb_half_incl_scan_u32(tot) - tot;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned lo = wd[j] & 0xFFFFu, hi = wd[j] >> 16;
            wd[j] = s | ((s + lo) << 16);
            s += lo + hi;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ((uint2 *)row)[8 * l5 + j] = make_uint2(wd[2 * j], wd[2 * j + 1]);
    }
    wave_lds_fence();
    KB_PH(8);
    // lane j < k of each half: the bin that holds approximate rank j d = the last bin with at most j d entries below it; flag
    // its window
    if (l5 < (int)a.k && T != 0u) {
        const unsigned m = (unsigned)l5 * a.d;
        unsigned lo = 0u, hi = 1023u;
#pragma unroll 1
        for (int it = 0; it < 10; ++it) {
            const unsigned mid = (lo + hi + 1u) >> 1;
            const bool le = (unsigned)cum16[mid] <= m;
            lo = le ? mid : lo;
            hi = le ? hi : mid - 1u;
        }
        const unsigned L = lo > w ? lo - w : 0u, H = lo + w < 1023u ? lo + w : 1023u;
        for (unsigned wi = L >> 5; wi <= (H >> 5); ++wi) {
            const unsigned first = wi == (L >> 5) ? (L & 31u) : 0u, lastb = wi == (H >> 5) ? (H & 31u) : 31u;
            atomicOr(&bitmap[wi], (0xFFFFFFFFu >> (31u - lastb)) & (0xFFFFFFFFu << first));
        }
    }
    wave_lds_fence();
    KB_PH(9);
    unsigned F = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool take = ((bitmap[bn[i] >> 5] >> (bn[i] & 31u)) & 1u) && (int)v[i] >= 0;
        const unsigned long long bl = __ballot(take);
        const unsigned pos = F + kb_half_prefix(bl, h);
        if (take && pos < (unsigned)KB_MAXFIN) fin[pos] = (unsigned long long)v[i];
        F += kb_half_count(bl, h);
    }
    wave_lds_fence();
    KB_PH(10);
    const unsigned Fo = (unsigned)__shfl_xor((int)F, 32);
    const unsigned Fmax = F > Fo ? F : Fo;
    if (Fmax > (unsigned)KB_MAXFIN) return 0xFFFFFFFFu;
    const float4 *xp4 = (const float4 *)a.xp;
    const float4 *q4 = (const float4 *)qbuf;
    const unsigned long long m44 = (1ull << 44) - 1ull;
    bool bound_broken = false;
    for (unsigned f0 = 0; f0 < Fmax; f0 += 32) {
        const unsigned f = f0 + (unsigned)l5;
        const unsigned ent = f < F ? (unsigned)fin[f] : 0u;
        const unsigned idx = ent & 0xFFFu;
        const size_t vc = a.room_row0 + idx;
        const float4 *pc = xp4 + (vc >> 4) * 256 + (vc & 15);
        float4 C[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) C[j] = pc[(j >> 2) * 64 + 16 * (j & 3)];
        const float sqj = a.sq[vc];
        // first rank of the finalist's run = entries below the first bin of the unbroken stretch of flagged bins it sits in
        unsigned bb = ((ent >> 12) + c0) >> sh;
        bb = ent <= low_ent ? 0u : (bb > 1023u ? 1023u : bb);
        int wi = (int)(bb >> 5);
        unsigned zeros = ~bitmap[wi] & (0xFFFFFFFFu >> (31u - (bb & 31u)));      // unflagged bins at or below bb in its word
        while (__ballot(zeros == 0u && wi > 0) != 0ull) {
            if (zeros == 0u && wi > 0) { --wi; zeros = ~bitmap[wi]; }
        }
        const unsigned b0 = zeros ? (unsigned)(wi * 32 + 32 - __builtin_clz(zeros)) : 0u;
        const unsigned long long tag = (unsigned long long)cum16[b0] << 44;
        // ascending-k fmaf chain from 0 (k = 16 i + 4 e + g), the exact kernel's arithmetic
        float z = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 qa = q4[4 * i], qb = q4[4 * i + 1], qc = q4[4 * i + 2], qd = q4[4 * i + 3];
            z = __fmaf_rn(qa.x, C[4 * i].x, z); z = __fmaf_rn(qa.y, C[4 * i + 1].x, z); z = __fmaf_rn(qa.z, C[4 * i + 2].x, z); z = __fmaf_rn(qa.w, C[4 * i + 3].x, z);
            z = __fmaf_rn(qb.x, C[4 * i].y, z); z = __fmaf_rn(qb.y, C[4 * i + 1].y, z); z = __fmaf_rn(qb.z, C[4 * i + 2].y, z); z = __fmaf_rn(qb.w, C[4 * i + 3].y, z);
            z = __fmaf_rn(qc.x, C[4 * i].z, z); z = __fmaf_rn(qc.y, C[4 * i + 1].z, z); z = __fmaf_rn(qc.z, C[4 * i + 2].z, z); z = __fmaf_rn(qc.w, C[4 * i + 3].z, z);
            z = __fmaf_rn(qd.x, C[4 * i].w, z); z = __fmaf_rn(qd.y, C[4 * i + 1].w, z); z = __fmaf_rn(qd.z, C[4 * i + 2].w, z); z = __fmaf_rn(qd.w, C[4 * i + 3].w, z);
        }
        // (|x_i|^2 + (-2 x_i.x_j)) + |x_j|^2, torch_edge.py:41-43, as in the exact kernel
        const float dd = __fadd_rn(__fmaf_rn(-2.0f, z, sq), sqj);
        if (f < F) fin[f] = tag | ((unsigned long long)key_of(dd) << 12) | (unsigned long long)idx;
        // The bound everything above rests on, L <= D <= L + 2 e_ij, assumes how the bf16 MFMA rounds internally (measured,
        // not documented).  Both sides are in registers here, so it is CHECKED for every finalist: the exact distance in key
        // units must lie in [key - 2, key + margin] (2 = the key's own rounding); a violation (or a NaN) sends the tile to the
        // exact path like any other failure, instead of ranking on a bound that did not hold.
        const float dk = __fmaf_rn(dd, inv_unit, KB_OFF), kf = (float)(ent >> 12);
        bound_broken |= f < F && !(dk >= kf - 2.0f && dk <= kf + (float)margin + 1.0f);
    }
    if (__ballot(bound_broken) != 0ull) return 0xFFFFFFFFu;
    wave_lds_fence();
    KB_PH(11);
    int32_t *out = a.out + (size_t)(h * KB_WAVES) * a.k;
    for (unsigned f0 = 0; f0 < Fmax; f0 += 32) {
        const unsigned f = f0 + (unsigned)l5;
        const unsigned long long x = f < F ? fin[f] : ~0ull;
        const unsigned long long xlo = x & ~m44;              // the smallest value of x's run
        // rank = first rank of the run + (finalists below x) - (finalists below the run)
        unsigned below = 0, below_run = 0;
        for (unsigned g = 0; g < Fmax; g += 2) {
            const unsigned long long y0 = g < F ? fin[g] : ~0ull, y1 = g + 1 < F ? fin[g + 1] : ~0ull;
            below += (y0 < x ? 1u : 0u) + (y1 < x ? 1u : 0u);
            below_run += (y0 < xlo ? 1u : 0u) + (y1 < xlo ? 1u : 0u);
        }
        const unsigned rank = (unsigned)(x >> 44) + below - below_run;
        const unsigned t = (rank * a.magic) >> 18;
        if (f < F && t * a.d == rank && t < a.k) out[t] = (int32_t)(x & 0xFFFull);
    }
    KB_PH(12);
    return F + Fo;
}

#ifndef KB_MIN_WAVES_EU
#define KB_MIN_WAVES_EU 4
#endif
__global__ __launch_bounds__(KB_WAVES * 64, KB_MIN_WAVES_EU) void knn_bf_kernel(KnnBfArgs a)
{
    constexpr int CAP = KB_CAP, ROW = KB_ROW;
    constexpr int PER_STEP = KB_WAVES * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    unsigned *ent = (unsigned *)kb_smem;                      // [KB_Q][ROW]
    float *thr_f = (float *)(ent + KB_Q * ROW);               // [KB_Q] admission threshold on acc
    unsigned *cnt = (unsigned *)(thr_f + KB_Q);               // [KB_Q]
    unsigned *tau = cnt + KB_Q;                               // [KB_Q] key threshold of the row
    float *rpar = (float *)(tau + KB_Q);                      // [KB_Q][KB_RP] row parameters
    float *scr = rpar + KB_Q * KB_RP + 192 * (threadIdx.x >> 6);     // [KB_WAVES][2][96] per wave and row: query features + bitmap
    unsigned *smax = (unsigned *)(rpar + KB_Q * KB_RP + 192 * KB_WAVES);   // [KB_WAVES]
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
#ifdef PSG_KF_TL
    unsigned long long tl0 = 0;
    int n_tl = 0;
    if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl0) :: "memory");
#endif
    __syncthreads();
    const unsigned mbits = wave_max_u32(smax[lane & (KB_WAVES - 1)]);
    // (bit patterns: NaN / inf / huge and zero / tiny rooms take the exact path)
    bool need_exact = !(mbits >= 0x20000000u && mbits <= 0x7C000000u);
    const float sqmaxf = __uint_as_float(mbits);
    if (tid < KB_Q) {
        // row parameters: the key unit is E_i(|x_i|^2) / 8 (a query far below the room's norms gets a floor, so that the
        // unit stays a normal number)
        const float sqi = a.sq[(size_t)tile * KB_Q + tid];
        const float sqe = fmaxf(sqi, sqmaxf * (1.0f / 4096.0f));
        const float ae = sqrtf(sqe), t0 = 2.0298f * ae;
        const float unit = 0.25f * (KB_A * ae * t0 + KB_B * t0 * t0 + KB_G * sqe);
        const float inv_unit = 1.0f / unit;
        float *rp = rpar + tid * KB_RP;
        rp[0] = sqi; rp[1] = sqrtf(sqi); rp[2] = unit; rp[3] = inv_unit;
        rp[4] = __fmaf_rn(sqi * (1.0f - KB_G), inv_unit, KB_OFF);      // key = round(acc * (-2 / u) + this)
        cnt[tid] = 0u;
        tau[tid] = KB_KEYMAX;
        // open row: everything up to the largest key is admitted (acc * m2 + ci < KEYMAX + 1, two units for the evaluation)
        thr_f[tid] = (rp[4] - (float)(KB_KEYMAX - 2u)) * (0.5f * unit);
    }
    __syncthreads();
    KB_TL(0);
    const float m2 = -2.0f * rpar[r * KB_RP + 3];
    const float ci23 = rpar[r * KB_RP + 4] + 8388608.0f;      // key = mantissa of acc * m2 + ci23
    unsigned long long st_fin = 0, st_cut = 0, st_ent = 0;
    KbRowArgs ra;
    ra.row_dw = (unsigned)(wave * ROW); ra.rp_dw = (unsigned)(rpar - (float *)kb_smem);
    ra.cnt_dw = (unsigned)(cnt - (unsigned *)kb_smem); ra.thr_dw = (unsigned)(thr_f - (float *)kb_smem);
    ra.tau_dw = (unsigned)(tau - (unsigned *)kb_smem); ra.KK = KK; ra.wave = wave;

    if (!need_exact) {
        const int n_steps = (tpr + KB_WAVES - 1) / KB_WAVES;
        const float grow = a.grow;
        unsigned next_prune = (unsigned)a.first_cut;
        // the augmented k-step's query side: 1, 1, 1 (x the three pieces of -(1 - B) |x_j|^2 / 2) and |x_i| rounded up to bf16
        // (x (A / 2) |x_j| rounded up) in elements 0..3 of the h = 0 lanes
        kb_u32x4 qa = kb_u32x4{0u, 0u, 0u, 0u};
        if (!h) {
            const unsigned au = __float_as_uint(rpar[r * KB_RP + 1]);
            qa = kb_u32x4{0x3F803F80u, 0x00003F80u | (((au + 0xFFFFu) >> 16) << 16), 0u, 0u};
        }
        for (int step = 0; step < n_steps; ++step) {
            const int t = step * KB_WAVES + wave;
            if (t < tpr) {
                kb_u32x4 fr[9];
#pragma unroll
                for (int s = 0; s < 9; ++s) fr[s] = a.bp[((tile0 + t) * 9 + s) * 64 + lane];
                kb_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const kb_bf16x8 ch = __builtin_bit_cast(kb_bf16x8, fr[s]), cl = __builtin_bit_cast(kb_bf16x8, fr[4 + s]);
                    const kb_bf16x8 qh = __builtin_bit_cast(kb_bf16x8, bq[s]), ql = __builtin_bit_cast(kb_bf16x8, bq[4 + s]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, qh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, ql, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl, qh, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kb_bf16x8, fr[8]), __builtin_bit_cast(kb_bf16x8, qa), acc, 0, 0, 0);
                // lane (query r, half h) holds the 16 candidates 32 t + (g & 3) + 8 (g >> 2) + 4 h
                const float th = thr_f[r];
                unsigned n = 0;
#pragma unroll
                for (int g = 0; g < 16; ++g) n += acc[g] >= th ? 1u : 0u;
                if (n) {
                    const unsigned base = atomicAdd(&cnt[r], n);
                    if (base + n <= (unsigned)CAP) {
                        // branch-free append: every pair builds its entry, the ones that fail the test store it to the
                        // thread's sink word (16 exec-masked blocks with two taken branches each cost more)
                        unsigned *dst = ent + r * ROW + base;
                        const unsigned cb = (unsigned)(32 * t + 4 * h);
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
            const unsigned seen = (unsigned)(step + 1) * (unsigned)PER_STEP;
            if (seen >= next_prune && step + 1 < n_steps) {
                const unsigned nn = (unsigned)(grow * (float)seen) / (unsigned)PER_STEP * (unsigned)PER_STEP;
                next_prune = nn > seen ? nn : seen + (unsigned)PER_STEP;
                __syncthreads();
#ifdef PSG_KF_TL
                KB_TL(n_tl < 3 ? 1 + 2 * n_tl : 5);
#endif
                // (every wave sees the same counters here: a row is only cut by waves that saw no overflow, and an
                // overflowing counter stays above CAP)
                if (__ballot(lane < KB_Q && cnt[lane & (KB_Q - 1)] > (unsigned)CAP) != 0ull) { need_exact = true; break; }
                {
                    const unsigned T0 = cnt[wave], T1 = cnt[wave + KB_WAVES];
                    ra.T[0] = T0 > (unsigned)a.LOW ? T0 : 0u;
                    ra.T[1] = T1 > (unsigned)a.LOW ? T1 : 0u;
                    const unsigned Tm = ra.T[0] > ra.T[1] ? ra.T[0] : ra.T[1];
                    if (Tm) {
                        if (Tm <= 512u) kb_cut_rows<16>(ra, lane); else kb_cut_rows<32>(ra, lane);
                        st_cut += (ra.T[0] ? 1u : 0u) + (ra.T[1] ? 1u : 0u);
                    }
                }
                __syncthreads();
#ifdef PSG_KF_TL
                KB_TL(n_tl < 3 ? 2 + 2 * n_tl : 6);
                ++n_tl;
#endif
            }
        }
        __syncthreads();
        KB_TL(8);
        // an overflowing row, or one that the key range left with fewer than KK candidates: exact path
        if (!need_exact) need_exact = __ballot(lane < KB_Q && (cnt[lane & (KB_Q - 1)] > (unsigned)CAP || cnt[lane & (KB_Q - 1)] < KK)) != 0ull;
        if (!need_exact) {
            KbFinalArgs fa;
            fa.xp = a.exact.xp; fa.sq = a.sq; fa.k = (unsigned)a.k; fa.d = (unsigned)a.d; fa.magic = a.magic;
            fa.room_row0 = room_row0; fa.qglob = (size_t)tile * KB_Q + wave;
            fa.out = a.out + fa.qglob * a.k;
            fa.scr_dw = (unsigned)(scr - (float *)kb_smem);
            ra.T[0] = cnt[wave]; ra.T[1] = cnt[wave + KB_WAVES];
            const unsigned Tm = ra.T[0] > ra.T[1] ? ra.T[0] : ra.T[1];
            const unsigned nf = Tm <= 256u ? kb_final_rows<8>(ra, fa, lane)
                              : (Tm <= 512u ? kb_final_rows<16>(ra, fa, lane) : kb_final_rows<32>(ra, fa, lane));
            if (nf == 0xFFFFFFFFu) { if (lane == 0) *fail = 1u; }
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
        if (lane == 0 && !need_exact) { atomicAdd(&acc3[0], (unsigned)st_fin); atomicAdd(&acc3[1], (unsigned)st_cut); atomicAdd(&acc3[2], (unsigned)st_ent); }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&a.stats[0], 1ull);
            if (need_exact) atomicAdd(&a.stats[1], 1ull);
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
    const size_t own = (size_t)KB_Q * KB_ROW * 4 + KB_Q * 12 + KB_Q * KB_RP * 4 + KB_WAVES * 192 * 4 + KB_WAVES * 4 + 16 + KB_WAVES * 64 * 4;
    return own > knn_fused_lds_bytes() ? own : knn_fused_lds_bytes();
}
