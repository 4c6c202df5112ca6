// Dilated kNN graph in feature space (C = 64) with a bf16 PREFILTER on the matrix cores: the exact fp32 distance - the one
// that decides the reference's neighbour order, ResGCN/gcn_lib/dense/torch_edge.py:32-59 - is evaluated only for the few
// candidates per row that can hold one of the wanted ranks; everything else is decided on an approximate distance with a
// PROVEN error bound.  Same results, bit for bit, as the exact kernel (psg_knn_fused.cuh), which stays in the same launch as
// the fallback for every tile whose bound cannot be kept.
//
// Approximation.  x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|.  One 32 x 32 tile of
//     acc = (hi_i.hi_j + hi_i.lo_j + lo_i.hi_j) - |x_j|^2 / 2
// is 13 v_mfma_f32_32x32x16_bf16 (K = 3 x 64 + one augmented step that multiplies three bf16 pieces of -|x_j|^2 / 2 by 1):
// 3/16 of the fp32 MFMA time of the exact kernel.  D~ = |x_i|^2 - 2 acc.  Error against the reference's fp32 distance D:
// dropped products (lo.lo and the split residues) <= 3 * 2^-18 |x_i||x_j|; fp32 accumulation of 13 instructions of 16
// exact products each, every addition rounded to nearest: <= 13 * 17 * 2^-24 S with S = |x_i||x_j| + |x_j|^2 / 2 <= 1.5 sqmax;
// the reference's own chain 64 * 2^-24 |x_i||x_j| and its two distance roundings 2^-22 sqmax.  In distance units (dot
// errors count twice): <= (2^-15.4 + 2^-14.6 + 2^-17 + 2^-21) sqmax < 2^-13.8 sqmax (sqmax = largest |x|^2 of the room).
// The kernel uses E = 2^-13 sqmax (1.8 x that bound); tools/bf16_split_probe.hip measured <= 2^-16.0 sqmax on hardware over eight data sets
// (random, offset, 1e6 dynamic range, near-cancelling pairs, 1e-15 and 1e15 magnitudes).  Rooms whose sqmax is not a normal
// number in [1e-19, 2.6e36] go to the exact path.
//
// Keys.  Distances are quantised linearly, key = round((D~ + 4 E) * 8 / E) < 2^18 (one FMA whose constant carries 2^23:
// the key is the result's mantissa), so
// |key - (D + 4 E) * 8 / E| <= EK = 10 units; a row entry is key << 12 | index (32 bits: twice the rows of the exact kernel
// in the same LDS).
//
// Stream (as in the exact kernel: 16 waves, shared row buffers, slots by one LDS atomic per lane and step, cuts on a
// data-independent schedule, no barrier in between) for 32 queries per workgroup; a candidate is admitted by one float
// compare acc >= th_i (a superset of key <= tau_i).  A cut finds thr with #(keys <= thr) >= KK (the histogram bin in which
// the cumulative count reaches KK) and keeps every entry with key <= tau = thr + 2 EK: the row always holds every candidate seen whose key is within 2 EK of the
// KK-th smallest key seen, hence - whatever the approximation did inside its bound - the true KK nearest.  A counter
// above CAP (adversarial orders, massive ties) sends the tile to the exact path.
//
// Final ranks.  Wanted: exact ranks 0, d, .., (k-1) d.  Entries are binned by key (1024 linear bins) and scanned as in the
// exact kernel; the element of exact rank m has its key within 2 EK of the m-th smallest key, so it lies within w bins
// (w = 2 EK / bin width + 2) of the bin that holds approximate rank m.  Those windows are flagged, overlapping windows
// merge into RUNS, the members of flagged bins ("finalists", a few per wanted rank) get their exact distance (ascending-k
// fmaf chain from the fp32 copy, the exact kernel's arithmetic) and rank themselves inside their run: everything below
// a run is smaller than, everything above it larger than each wanted element in it, so first rank of the run + the number
// of smaller finalists of the same run IS the exact rank.  More than 256 finalists, or windows wider than 15 bins with more
// than 256 entries in the row, send the tile to the exact path.
#pragma once

constexpr int KB_WAVES = 16;
constexpr int KB_Q = 32;                      // query rows per workgroup (one 32-column MFMA tile)
constexpr int KB_CAP = 1024;                  // entries per row buffer
constexpr int KB_ROW = KB_CAP + 2;            // dwords between rows: 8-byte aligned, skewed over the LDS banks
constexpr int KB_NPL = KB_CAP / 64;
constexpr float KB_UNITS = 65536.0f;          // key units per sqmax: 8 / E with E = 2^-13 sqmax
constexpr float KB_OFF = 32.0f;               // key offset (4 E): slightly negative distances stay un-clamped
constexpr unsigned KB_EK = 10u;               // |key - exact distance in key units| <= 8 (E) + evaluation and truncation
constexpr unsigned KB_MARGIN = 2u * KB_EK;
constexpr int KB_MAXFIN = 256;

typedef __bf16 kb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float kb_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned kb_u32x4 __attribute__((ext_vector_type(4)));     // one 16-byte operand fragment

// Diagnostic build (EXTRA=-DPSG_KF_TL, tools/knn_timeline.py): thread 0 adds the time since kernel start at the workgroup's
// barrier exits into the exact kernel's g_kf_tl table (same slots: 0 start, 1-6 first three cuts, 8 stream end, 9 wave 0
// ranked, 10 all ranked)
#ifdef PSG_KF_TL
#define KB_TL(idx) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    atomicAdd(&g_kf_tl[(idx)], t_ - tl0); atomicAdd(&g_kf_tl[16 + (idx)], 1ull); } } while (0)
#else
#define KB_TL(idx)
#endif

#ifdef PSG_KF_TL
__device__ unsigned long long g_kb_ph[32];      // phase cycle sums of the final ranking / cuts (wave 0 of every workgroup)
#define KB_PH_BEGIN unsigned long long ph_t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_t0_) :: "memory")
#define KB_PH(idx) do { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (threadIdx.x == 0) { atomicAdd(&g_kb_ph[(idx)], t_ - ph_t0_); atomicAdd(&g_kb_ph[16 + (idx)], 1ull); } ph_t0_ = t_; } while (0)
#else
#define KB_PH_BEGIN
#define KB_PH(idx)
#endif

struct KnnBfArgs {
    const kb_u32x4 *bp;   // [rows / 32][9][64] fragments (psg_knn_ops.cuh)
    const float *sq;   // [rows]
    int32_t *out;      // [rows][k]
    int N;             // points per room, multiple of 32
    int k, d, KK;
    unsigned magic;    // ceil(2^18 / d)
    int LOW;           // a cut shortens the rows that hold more than LOW entries
    int first_cut;     // candidates seen at the first cut (a multiple of 512)
    float grow;        // a cut at n candidates seen is followed by one at grow * n
    KnnFusedArgs exact;            // the exact path's arguments (xp, sq, out, ...)
    unsigned long long *stats;     // optional [8]: tiles, fallback tiles, rows, finalists, cut events, entries at the end
};

__device__ __forceinline__ unsigned kb_wave_excl_scan_max(unsigned v, int lane)
{
    unsigned o;
    o = dpp_get<0x111, 0xF>(v); v = o > v ? o : v;
    o = dpp_get<0x112, 0xF>(v); v = o > v ? o : v;
    o = dpp_get<0x114, 0xF>(v); v = o > v ? o : v;
    o = dpp_get<0x118, 0xF>(v); v = o > v ? o : v;
    o = dpp_get<0x142, 0xA>(v); v = o > v ? o : v;
    o = dpp_get<0x143, 0xC>(v); v = o > v ? o : v;
    const unsigned up = (unsigned)__shfl_up((int)v, 1);
    return lane ? up : 0u;
}

// ---- cuts and final ranking of a row.  Both work on the row's entries in registers (element i of lane l = entry 64 i + l,
// slots beyond T hold the sentinel 0xFFFFFFFF: real entries are below 2^31) and use the row's own LDS as scratch.
// (Not inlined: with the cuts, the final ranking and the exact path inlined the kernel spilled inside its stream loop.  LDS
// locations travel as dword offsets into the workgroup's dynamic LDS, so that every access stays a ds_ instruction: a
// pointer parameter would be a generic pointer and turn them into flat_ accesses.)

// Cut R rows (one wave; R = 2: the wave's two rows side by side, two independent dependency chains for the scheduler: these
// phases are bound by DPP / LDS latency, not by issue).  Per row: the first entry value above the smallest key's 2 EK
// neighbourhood (the query itself sits far below the rest in high dimensions) starts 256 linear bins of 32-bit counters,
// everything below goes to bin 0; the bin in which the cumulative count reaches KK gives thr with #(entries <= thr) >= KK;
// every entry with key <= tauk = key(thr) + 2 EK (+ 2 for the float bin edges) is kept, compacted in place.  Lane 0 stores
// the new count and the admission threshold on acc of each row.
struct KbCutArgs { unsigned row_dw, T[2], KK, cnt_dw, thr_dw; float ci[2], half_unit; };

template <int NV, int R>
__device__ __noinline__ void kb_cut_rows(const KbCutArgs c, int lane)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    unsigned *row[R];
    unsigned v[R][NV], mn[R], mx1[R];
    KB_PH_BEGIN;
#pragma unroll
    for (int rw = 0; rw < R; ++rw) {
        row[rw] = (unsigned *)kb_smem + c.row_dw + rw * KB_WAVES * KB_ROW;
        mn[rw] = 0xFFFFFFFFu; mx1[rw] = 0u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned e = (unsigned)(i * 64 + lane);
            const unsigned x = row[rw][e];
            v[rw][i] = e < c.T[rw] ? x : 0xFFFFFFFFu;
            mn[rw] = v[rw][i] < mn[rw] ? v[rw][i] : mn[rw];
            mx1[rw] = v[rw][i] + 1u > mx1[rw] ? v[rw][i] + 1u : mx1[rw];          // (the sentinel wraps to 0)
        }
    }
    wave_lds_fence();
    KB_PH(0);
    unsigned lowE[R], mn2[R];
#pragma unroll
    for (int rw = 0; rw < R; ++rw) {
        mn[rw] = ~wave_max_u32(~mn[rw]);
        lowE[rw] = (((mn[rw] >> 12) + KB_MARGIN) << 12) | 0xFFFu;               // entries up to here: bin 0
        mn2[rw] = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NV; ++i) mn2[rw] = (v[rw][i] > lowE[rw] && v[rw][i] < mn2[rw]) ? v[rw][i] : mn2[rw];
    }
    float scale[R], inv_scale[R];
#pragma unroll
    for (int rw = 0; rw < R; ++rw) {
        const unsigned mx = wave_max_u32(mx1[rw]) - 1u;
        mn2[rw] = ~wave_max_u32(~mn2[rw]);                                      // 0xFFFFFFFF (sentinel or none): one bin
        mn2[rw] = mn2[rw] > mx ? mx : mn2[rw];
        const float span = (float)(mx - mn2[rw] + 1u);
        scale[rw] = 254.0f / span; inv_scale[rw] = span * (1.0f / 254.0f);
        // 256 counters; the lane owns bins 4 lane .. 4 lane + 3
        ((uint2 *)row[rw])[2 * lane] = make_uint2(0u, 0u);
        ((uint2 *)row[rw])[2 * lane + 1] = make_uint2(0u, 0u);
    }
    wave_lds_fence();
    KB_PH(1);
#pragma unroll
    for (int rw = 0; rw < R; ++rw)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned bb = 1u + (unsigned)((float)(v[rw][i] - mn2[rw]) * scale[rw]);   // (sentinels land in bin 255 and are taken out below)
            bb = v[rw][i] <= lowE[rw] ? 0u : (bb > 255u ? 255u : bb);
            atomicAdd(&row[rw][bb], 1u);
        }
    wave_lds_fence();
    KB_PH(2);
    unsigned keepmax[R];
#pragma unroll
    for (int rw = 0; rw < R; ++rw) {
        const uint2 ca = ((const uint2 *)row[rw])[2 * lane], cb = ((const uint2 *)row[rw])[2 * lane + 1];
        const unsigned c3 = cb.y - (lane == 63 ? (unsigned)(NV * 64) - c.T[rw] : 0u);
        const unsigned tot = ca.x + ca.y + cb.x + c3;
        const unsigned incl = wave_incl_scan_u32(tot), start = incl - tot;
        const unsigned s0 = start + ca.x, s1 = s0 + ca.y, s2 = s1 + cb.x;
        const unsigned mybin = (unsigned)(4 * lane) + (s0 < c.KK ? 1u : 0u) + (s1 < c.KK ? 1u : 0u) + (s2 < c.KK ? 1u : 0u);
        const unsigned long long own = __ballot(start < c.KK && c.KK <= incl);          // exactly one lane (KK <= T)
        const unsigned binB = (unsigned)__builtin_amdgcn_readlane((int)mybin, (int)__builtin_ctzll(own));
        // an entry of a bin <= binB (>= 1) has 1 + (v - mn2) * scale < binB + 1 in float: v - mn2 < binB / scale * (1 + 2^-22),
        // less than one key above; key(thr) + 2 covers that and the floor of the shift
        const unsigned X = binB ? mn2[rw] + (unsigned)((float)binB * inv_scale[rw]) : lowE[rw];
        unsigned tauk = (X >> 12) + 2u + KB_MARGIN;
        tauk = tauk > 0xFFFFEu ? 0xFFFFEu : tauk;
        keepmax[rw] = (tauk << 12) | 0xFFFu;
        // admit iff key <= tauk, i.e. acc * m2 + ci < tauk + 1; two more units cover the float evaluation
        if (lane == 0) ((float *)kb_smem)[c.thr_dw + rw * KB_WAVES] = (c.ci[rw] - (float)(tauk + 3u)) * c.half_unit;
    }
    KB_PH(3);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned base[R];
#pragma unroll
    for (int rw = 0; rw < R; ++rw) base[rw] = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int rw = 0; rw < R; ++rw) {
            const bool keep = v[rw][i] <= keepmax[rw];                  // (sentinels are above any keepmax)
            const unsigned long long bl = __ballot(keep);
            if (keep) row[rw][base[rw] + __popcll(bl & lt_mask)] = v[rw][i];
            base[rw] += (unsigned)__popcll(bl);
        }
    if (lane == 0) {
#pragma unroll
        for (int rw = 0; rw < R; ++rw) ((unsigned *)kb_smem)[c.cnt_dw + rw * KB_WAVES] = base[rw];
    }
    KB_PH(4);
}

// Exact ranks 0, d, .., (k-1) d of R rows (the wave's two, side by side like the cuts) from their T >= KK approximate
// entries (see the header).  Row LDS: 1024 bins of 16 bits (counts, then first ranks) + 256 finalists; scr = 96 words of
// wave-private LDS per row (the query's 64 features, 32 words of flagged-bin bitmap).  Bin 0 takes every key within 2 EK of
// the smallest, bins 1.. have a power-of-two width over the keys above that: exact integer edges, so a window is
// w = ceil(2 EK / width) bins to each side.  Returns the number of finalists, or 0xFFFFFFFF when a row must take the exact
// path.
struct KbFinalArgs {          // by value: a reference to the kernel's argument struct would force it (and every pointer in it) through the stack
    const float *xp, *sq;
    int32_t *out;             // the first row's output; the second row's is 16 k further
    size_t room_row0, qglob;  // global row of the room's first point and of the first row's query (second: + 16)
    unsigned row_dw, scr_dw, T[2], k, d, magic;
    float sqi[2];
};

template <int NV, int R>
__device__ __noinline__ unsigned kb_final_rows(const KbFinalArgs a, int lane)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    unsigned *row[R], *bitmap[R];
    float *qbuf[R];
    unsigned v[R][NV], bn[R][NV], mn[R], mx1[R];
    KB_PH_BEGIN;
#pragma unroll
    for (int rw = 0; rw < R; ++rw) {
        row[rw] = (unsigned *)kb_smem + a.row_dw + rw * KB_WAVES * KB_ROW;
        qbuf[rw] = (float *)kb_smem + a.scr_dw + rw * 96;
        bitmap[rw] = (unsigned *)kb_smem + a.scr_dw + rw * 96 + 64;
        mn[rw] = 0xFFFFFFFFu; mx1[rw] = 0u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned e = (unsigned)(i * 64 + lane);
            const unsigned x = row[rw][e];
            v[rw][i] = e < a.T[rw] ? x : 0xFFFFFFFFu;
            mn[rw] = v[rw][i] < mn[rw] ? v[rw][i] : mn[rw];
            mx1[rw] = v[rw][i] + 1u > mx1[rw] ? v[rw][i] + 1u : mx1[rw];
        }
        // the query's 64 features in k order for the exact chains (feature k = 16 i + 4 e + g: element e of float4 [i][g])
        const size_t qg = a.qglob + (size_t)(rw * KB_WAVES);
        const float *xq = a.xp + ((qg >> 4) * 256 + (qg & 15)) * 4;
        qbuf[rw][lane] = xq[((lane >> 4) * 64 + 16 * (lane & 3)) * 4 + ((lane >> 2) & 3)];
        if (lane < 32) bitmap[rw][lane] = 0u;
    }
    wave_lds_fence();
    KB_PH(5);
    unsigned kmx[R], lowmax[R], k2[R];
#pragma unroll
    for (int rw = 0; rw < R; ++rw) {
        lowmax[rw] = ((~wave_max_u32(~mn[rw])) >> 12) + KB_MARGIN;
        k2[rw] = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned kq = v[rw][i] >> 12;
            k2[rw] = (kq > lowmax[rw] && kq < k2[rw]) ? kq : k2[rw];          // (a sentinel's key 0xFFFFF is above every real key)
        }
    }
    unsigned sh[R], w[R], c0[R], low_ent[R];
#pragma unroll
    for (int rw = 0; rw < R; ++rw) {
        kmx[rw] = (wave_max_u32(mx1[rw]) - 1u) >> 12;
        k2[rw] = ~wave_max_u32(~k2[rw]);
        const unsigned range = (k2[rw] <= kmx[rw]) ? kmx[rw] - k2[rw] : 0u;
        unsigned s = range > 1022u ? (unsigned)(32 - __builtin_clz(range)) - 10u : 0u;
        if ((range >> s) > 1022u) ++s;                              // bins 1 .. 1 + (range >> s) <= 1023
        sh[rw] = s;
        w[rw] = (KB_MARGIN + (1u << s) - 1u) >> s;                  // window half width in bins (<= 2 EK = 20)
        c0[rw] = ((1u << s) - k2[rw]) << 12;                        // bin = (v + c0) >> (12 + s) above lowmax
        low_ent[rw] = (lowmax[rw] << 12) | 0xFFFu;
#pragma unroll
        for (int j = 0; j < 4; ++j) ((uint2 *)row[rw])[4 * lane + j] = make_uint2(0u, 0u);
    }
    wave_lds_fence();
    KB_PH(6);
#pragma unroll
    for (int rw = 0; rw < R; ++rw)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned bb = (v[rw][i] + c0[rw]) >> (12u + sh[rw]);
            bb = v[rw][i] <= low_ent[rw] ? 0u : (bb > 1023u ? 1023u : bb);   // (sentinels: bin 1023, taken out below)
            bn[rw][i] = bb;
            atomicAdd(&row[rw][bb >> 1], 1u + (bb & 1u) * 0xFFFFu);
        }
    wave_lds_fence();
    KB_PH(7);
    // the lane's own 16 bins (16 lane .. 16 lane + 15) -> the number of entries below each bin, written back in place
#pragma unroll
    for (int rw = 0; rw < R; ++rw) {
        unsigned wd[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint2 t = ((const uint2 *)row[rw])[4 * lane + j]; wd[2 * j] = t.x; wd[2 * j + 1] = t.y; }
        if (lane == 63) wd[7] -= ((unsigned)(NV * 64) - a.T[rw]) << 16;
        unsigned tot = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) tot += (wd[j] & 0xFFFFu) + (wd[j] >> 16);
        unsigned s = wave_incl_scan_u32(tot) - tot;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned lo = wd[j] & 0xFFFFu, hi = wd[j] >> 16;
            wd[j] = s | ((s + lo) << 16);
            s += lo + hi;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ((uint2 *)row[rw])[4 * lane + j] = make_uint2(wd[2 * j], wd[2 * j + 1]);
    }
    wave_lds_fence();
    KB_PH(8);
    // lane j < k: the bin that holds approximate rank j d = the last bin with at most j d entries below it; flag its window
    if (lane < (int)a.k) {
        const unsigned m = (unsigned)lane * a.d;
        unsigned lo[R], hi[R];
#pragma unroll
        for (int rw = 0; rw < R; ++rw) { lo[rw] = 0u; hi[rw] = 1023u; }
#pragma unroll 1
        for (int it = 0; it < 10; ++it)
#pragma unroll
            for (int rw = 0; rw < R; ++rw) {
                const unsigned mid = (lo[rw] + hi[rw] + 1u) >> 1;
                const bool le = (unsigned)((const unsigned short *)row[rw])[mid] <= m;
                lo[rw] = le ? mid : lo[rw];
                hi[rw] = le ? hi[rw] : mid - 1u;
            }
#pragma unroll
        for (int rw = 0; rw < R; ++rw) {
            const unsigned L = lo[rw] > w[rw] ? lo[rw] - w[rw] : 0u, H = lo[rw] + w[rw] < 1023u ? lo[rw] + w[rw] : 1023u;
            for (unsigned wi = L >> 5; wi <= (H >> 5); ++wi) {
                const unsigned first = wi == (L >> 5) ? (L & 31u) : 0u, lastb = wi == (H >> 5) ? (H & 31u) : 31u;
                atomicOr(&bitmap[rw][wi], (0xFFFFFFFFu >> (31u - lastb)) & (0xFFFFFFFFu << first));
            }
        }
    }
    wave_lds_fence();
    KB_PH(9);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned F[R];
#pragma unroll
    for (int rw = 0; rw < R; ++rw) F[rw] = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int rw = 0; rw < R; ++rw) {
            unsigned long long *fin = (unsigned long long *)(row[rw] + 512);
            const bool take = ((bitmap[rw][bn[rw][i] >> 5] >> (bn[rw][i] & 31u)) & 1u) && (int)v[rw][i] >= 0;
            const unsigned long long bl = __ballot(take);
            const unsigned pos = F[rw] + (unsigned)__popcll(bl & lt_mask);
            if (take && pos < (unsigned)KB_MAXFIN) fin[pos] = (unsigned long long)v[rw][i];
            F[rw] += (unsigned)__popcll(bl);
        }
    wave_lds_fence();
    KB_PH(10);
    unsigned Fmax = 0, Fsum = 0;
#pragma unroll
    for (int rw = 0; rw < R; ++rw) { Fmax = F[rw] > Fmax ? F[rw] : Fmax; Fsum += F[rw]; }
    if (Fmax > (unsigned)KB_MAXFIN) return 0xFFFFFFFFu;
    const float4 *xp4 = (const float4 *)a.xp;
    const unsigned long long m44 = (1ull << 44) - 1ull;
    for (unsigned f0 = 0; f0 < Fmax; f0 += 64) {
        const unsigned f = f0 + (unsigned)lane;
#pragma unroll
        for (int rw = 0; rw < R; ++rw) {
            unsigned long long *fin = (unsigned long long *)(row[rw] + 512);
            const float4 *q4 = (const float4 *)qbuf[rw];
            const unsigned ent = f < F[rw] ? (unsigned)fin[f] : 0u;
            const unsigned idx = ent & 0xFFFu;
            const size_t vc = a.room_row0 + idx;
            const float4 *pc = xp4 + (vc >> 4) * 256 + (vc & 15);
            float4 C[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) C[j] = pc[(j >> 2) * 64 + 16 * (j & 3)];
            const float sqj = a.sq[vc];
            // first rank of the finalist's run = entries below the first bin of the unbroken stretch of flagged bins it sits in
            unsigned bb = (ent + c0[rw]) >> (12u + sh[rw]);
            bb = ent <= low_ent[rw] ? 0u : (bb > 1023u ? 1023u : bb);
            int wi = (int)(bb >> 5);
            unsigned zeros = ~bitmap[rw][wi] & (0xFFFFFFFFu >> (31u - (bb & 31u)));      // unflagged bins at or below bb in its word
            while (__ballot(zeros == 0u && wi > 0) != 0ull) {
                if (zeros == 0u && wi > 0) { --wi; zeros = ~bitmap[rw][wi]; }
            }
            const unsigned b0 = zeros ? (unsigned)(wi * 32 + 32 - __builtin_clz(zeros)) : 0u;
            const unsigned long long tag = (unsigned long long)((const unsigned short *)row[rw])[b0] << 44;
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
            const float dd = __fadd_rn(__fmaf_rn(-2.0f, z, a.sqi[rw]), sqj);
            if (f < F[rw]) fin[f] = tag | ((unsigned long long)key_of(dd) << 12) | (unsigned long long)idx;
        }
    }
    wave_lds_fence();
    KB_PH(11);
    const unsigned d = a.d;
    for (unsigned f0 = 0; f0 < Fmax; f0 += 64) {
        const unsigned f = f0 + (unsigned)lane;
        unsigned long long x[R], xlo[R];
        unsigned below[R], below_run[R];
#pragma unroll
        for (int rw = 0; rw < R; ++rw) {
            const unsigned long long *fin = (const unsigned long long *)(row[rw] + 512);
            x[rw] = f < F[rw] ? fin[f] : ~0ull;
            xlo[rw] = x[rw] & ~m44;                             // the smallest value of x's run
            below[rw] = 0u; below_run[rw] = 0u;
        }
        // rank = first rank of the run + (finalists below x) - (finalists below the run)
        for (unsigned g = 0; g < Fmax; g += 2)
#pragma unroll
            for (int rw = 0; rw < R; ++rw) {
                const unsigned long long *fin = (const unsigned long long *)(row[rw] + 512);
                const unsigned long long y0 = g < F[rw] ? fin[g] : ~0ull, y1 = g + 1 < F[rw] ? fin[g + 1] : ~0ull;
                below[rw] += (y0 < x[rw] ? 1u : 0u) + (y1 < x[rw] ? 1u : 0u);
                below_run[rw] += (y0 < xlo[rw] ? 1u : 0u) + (y1 < xlo[rw] ? 1u : 0u);
            }
#pragma unroll
        for (int rw = 0; rw < R; ++rw) {
            const unsigned rank = (unsigned)(x[rw] >> 44) + below[rw] - below_run[rw];
            const unsigned t = (rank * a.magic) >> 18;
            if (f < F[rw] && t * d == rank && t < a.k) a.out[(size_t)(rw * KB_WAVES) * a.k + t] = (int32_t)(x[rw] & 0xFFFull);
        }
    }
    KB_PH(12);
    return Fsum;
}

__global__ __launch_bounds__(KB_WAVES * 64) void knn_bf_kernel(KnnBfArgs a)
{
    constexpr int CAP = KB_CAP, ROW = KB_ROW;
    constexpr int PER_STEP = KB_WAVES * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    unsigned *ent = (unsigned *)kb_smem;                      // [KB_Q][ROW]
    float *thr_f = (float *)(ent + KB_Q * ROW);               // [KB_Q] admission threshold on acc
    unsigned *cnt = (unsigned *)(thr_f + KB_Q);               // [KB_Q]
    float *csq = (float *)(cnt + KB_Q);                       // [KB_Q] |x_i|^2 of the tile's queries
    float *scr = csq + KB_Q + 192 * (threadIdx.x >> 6);       // [KB_WAVES][2][96] per wave and row: the query's features + flagged-bin bitmap (final ranking)
    unsigned *smax = (unsigned *)(csq + KB_Q + 192 * KB_WAVES);  // [KB_WAVES]
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

    if (tid < KB_Q) { thr_f[tid] = -INFINITY; cnt[tid] = 0u; csq[tid] = a.sq[(size_t)tile * KB_Q + tid]; }
    if (tid == 0) *fail = 0u;
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
    // the tile's queries as the B operand: hi (k-steps 0..3) and lo (4..7) fragments
    kb_u32x4 bq[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) bq[s] = a.bp[((size_t)tile * 9 + s) * 64 + lane];
    const float sqi = a.sq[(size_t)tile * KB_Q + r];
#ifdef PSG_KF_TL
    unsigned long long tl0 = 0;
    int n_tl = 0;
    if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl0) :: "memory");
#endif
    __syncthreads();
    KB_TL(0);
    const unsigned mbits = wave_max_u32(smax[lane & (KB_WAVES - 1)]);
    // (bit patterns: NaN / inf / huge and zero / tiny rooms take the exact path)
    bool need_exact = !(mbits >= 0x20000000u && mbits <= 0x7C000000u);
    const float sqmaxf = __uint_as_float(mbits);
    const float inv_unit = KB_UNITS / sqmaxf, half_unit = sqmaxf * (0.5f / KB_UNITS);
    const float m2 = -2.0f * inv_unit;
    const float ci23 = __fmaf_rn(sqi, inv_unit, KB_OFF) + 8388608.0f;   // key = round(acc * m2 + ci): mantissa of acc * m2 + ci23
    unsigned long long st_fin = 0, st_cut = 0, st_ent = 0;

    if (!need_exact) {
        const int n_steps = (tpr + KB_WAVES - 1) / KB_WAVES;
        const float grow = a.grow;
        unsigned next_prune = (unsigned)a.first_cut;
        // the augmented k-step's query side: 1, 1, 1 in elements 0..2 of the h = 0 lanes
        const kb_u32x4 qa = h ? kb_u32x4{0u, 0u, 0u, 0u} : kb_u32x4{0x3F803F80u, 0x00003F80u, 0u, 0u};
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
                    // the wave's two rows (wave, wave + 16) side by side when both are due, else the one that is
                    const unsigned T0 = cnt[wave], T1 = cnt[wave + KB_WAVES];
                    const bool c0 = T0 > (unsigned)a.LOW, c1 = T1 > (unsigned)a.LOW;
                    KbCutArgs ca;
                    ca.KK = KK; ca.half_unit = half_unit;
                    const int first = c0 ? 0 : 1;
                    ca.row_dw = (unsigned)((wave + first * KB_WAVES) * ROW);
                    ca.cnt_dw = (unsigned)(cnt - (unsigned *)kb_smem) + (unsigned)(wave + first * KB_WAVES);
                    ca.thr_dw = (unsigned)(thr_f - (float *)kb_smem) + (unsigned)(wave + first * KB_WAVES);
                    ca.T[0] = c0 ? T0 : T1; ca.T[1] = T1;
                    ca.ci[0] = __fmaf_rn(csq[wave + first * KB_WAVES], inv_unit, KB_OFF);
                    ca.ci[1] = __fmaf_rn(csq[wave + KB_WAVES], inv_unit, KB_OFF);
                    const unsigned Tm = c0 && c1 ? (T0 > T1 ? T0 : T1) : ca.T[0];
                    if (c0 && c1) { if (Tm <= 512u) kb_cut_rows<8, 2>(ca, lane); else kb_cut_rows<KB_NPL, 2>(ca, lane); st_cut += 2; }
                    else if (c0 || c1) { if (Tm <= 512u) kb_cut_rows<8, 1>(ca, lane); else kb_cut_rows<KB_NPL, 1>(ca, lane); st_cut += 1; }
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
        if (!need_exact) need_exact = __ballot(lane < KB_Q && cnt[lane & (KB_Q - 1)] > (unsigned)CAP) != 0ull;
        if (!need_exact) {
            bool ok = true;
            {
                KbFinalArgs fa;
                fa.xp = a.exact.xp; fa.sq = a.sq; fa.k = (unsigned)a.k; fa.d = (unsigned)a.d; fa.magic = a.magic;
                fa.room_row0 = room_row0; fa.qglob = (size_t)tile * KB_Q + wave;
                fa.out = a.out + fa.qglob * a.k;
                fa.row_dw = (unsigned)(wave * ROW); fa.scr_dw = (unsigned)(scr - (float *)kb_smem);
                fa.T[0] = cnt[wave]; fa.T[1] = cnt[wave + KB_WAVES];
                fa.sqi[0] = csq[wave]; fa.sqi[1] = csq[wave + KB_WAVES];
                const unsigned Tm = fa.T[0] > fa.T[1] ? fa.T[0] : fa.T[1];
                const unsigned nf = Tm <= 256u ? kb_final_rows<4, 2>(fa, lane)
                                  : (Tm <= 512u ? kb_final_rows<8, 2>(fa, lane) : kb_final_rows<KB_NPL, 2>(fa, lane));
                ok = nf != 0xFFFFFFFFu;
                st_fin += nf; st_ent += fa.T[0] + fa.T[1];
            }
            if (!ok && lane == 0) *fail = 1u;
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
    const size_t own = (size_t)KB_Q * KB_ROW * 4 + KB_Q * 12 + KB_WAVES * 192 * 4 + KB_WAVES * 4 + 16 + KB_WAVES * 64 * 4;
    return own > knn_fused_lds_bytes() ? own : knn_fused_lds_bytes();
}
