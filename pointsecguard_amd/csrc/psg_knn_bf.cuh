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

// ---- a row's entries binned by key: the machinery shared by the cuts and the final ranking.  1024 bins of 16 bits (512
// words of the row's own LDS, which is scratch once the entries are in registers).  Bin 0 takes every key within 2 EK of the
// smallest (the query itself, far below the rest in high dimensions: a linear map from the minimum would leave most bins
// empty), bins 1..1023 are linear between the first key above that and the largest: a monotone map.
struct KbBins { unsigned lowmax, k2; float scale; };

template <int NV>
__device__ __forceinline__ KbBins kb_bins_of_row(const unsigned (&v)[NV], unsigned T, int lane)
{
    unsigned kmn = 0xFFFFFFFFu, kmx = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool valid = (unsigned)(i * 64 + lane) < T;
        const unsigned kq = v[i] >> 12;
        kmn = (valid && kq < kmn) ? kq : kmn;
        kmx = (valid && kq > kmx) ? kq : kmx;
    }
    kmn = ~wave_max_u32(~kmn);
    kmx = wave_max_u32(kmx);
    KbBins B;
    B.lowmax = kmn + KB_MARGIN;
    unsigned k2 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool valid = (unsigned)(i * 64 + lane) < T;
        const unsigned kq = v[i] >> 12;
        k2 = (valid && kq > B.lowmax && kq < k2) ? kq : k2;
    }
    k2 = ~wave_max_u32(~k2);
    B.k2 = k2 == 0xFFFFFFFFu ? B.lowmax + 1u : k2;
    const unsigned range = kmx > B.k2 ? kmx - B.k2 : 1u;
    B.scale = 1022.0f / (float)range;
    return B;
}
__device__ __forceinline__ unsigned kb_bin(const KbBins &B, unsigned ent)
{
    const unsigned kq = ent >> 12;
    const unsigned b = 1u + (unsigned)((float)(kq - B.k2) * B.scale);       // (kq < k2 only when kq <= lowmax)
    return kq <= B.lowmax ? 0u : (b > 1023u ? 1023u : b);
}

// counts of the 1024 bins -> hist; the lane's own 16 bins (16 lane .. 16 lane + 15) come back in wd, `start` = number of
// entries in the bins before them; bn[i] = bin of entry i
template <int NV>
__device__ __forceinline__ void kb_histogram(const unsigned (&v)[NV], unsigned T, const KbBins &B, unsigned *hist, int lane,
                                             unsigned (&bn)[NV], unsigned (&wd)[8], unsigned &start)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) ((uint2 *)hist)[4 * lane + j] = make_uint2(0u, 0u);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        bn[i] = kb_bin(B, v[i]);
        if ((unsigned)(i * 64 + lane) < T) atomicAdd(&hist[bn[i] >> 1], 1u << (16 * (bn[i] & 1u)));
    }
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint2 t = ((const uint2 *)hist)[4 * lane + j]; wd[2 * j] = t.x; wd[2 * j + 1] = t.y; }
    unsigned tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += (wd[j] & 0xFFFFu) + (wd[j] >> 16);
    start = wave_incl_scan_u32(tot) - tot;
}

// Cut a row of T entries: the bin in which the cumulative count reaches KK gives thr = the largest key up to that bin
// (KK <= #(keys <= thr) <= KK + that bin's population); every entry with key <= tauk = thr + 2 EK is kept, compacted in place.
// One wave; returns the number kept.
// (not inlined: with the cuts, the final ranking and the exact path inlined the kernel spilled inside its stream loop.  LDS
// locations travel as dword offsets into the workgroup's dynamic LDS, so that every access stays a ds_ instruction: a
// pointer parameter would be a generic pointer and turn them into flat_ accesses)
template <int NV>
__device__ __noinline__ unsigned kb_cut_row(unsigned row_dw, unsigned T, unsigned KK, int lane, unsigned &tauk_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    unsigned *row = (unsigned *)kb_smem + row_dw;
    unsigned v[NV], bn[NV], wd[8], start;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned e = (unsigned)(i * 64 + lane);
        v[i] = e < T ? row[e] : 0xFFFFFFFFu;
    }
    wave_lds_fence();
    const KbBins B = kb_bins_of_row<NV>(v, T, lane);
    kb_histogram<NV>(v, T, B, row, lane, bn, wd, start);
    unsigned s = start, tb = 0xFFFFu;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const unsigned h = (wd[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
        if (s < KK && KK <= s + h) tb = (unsigned)(16 * lane + t);
        s += h;
    }
    const unsigned binB = ~wave_max_u32(~tb);                 // exactly one lane holds the crossing bin (KK <= T)
    unsigned kth = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned kq = v[i] >> 12;
        kth = ((unsigned)(i * 64 + lane) < T && bn[i] <= binB && kq > kth) ? kq : kth;
    }
    kth = wave_max_u32(kth);
    unsigned tauk = kth + KB_MARGIN;
    tauk = tauk > 0xFFFFEu ? 0xFFFFEu : tauk;
    const unsigned keepmax = (tauk << 12) | 0xFFFu;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned base = 0;
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool keep = (unsigned)(i * 64 + lane) < T && v[i] <= keepmax;
        const unsigned long long bl = __ballot(keep);
        if (keep) row[base + __popcll(bl & lt_mask)] = v[i];
        base += (unsigned)__popcll(bl);
    }
    tauk_out = tauk;
    return base;
}

// Exact ranks 0, d, .., (k-1) d of one row from its T >= KK approximate entries (see the header).  `row` = the row's LDS
// (entries on entry, scratch afterwards: 512 words of 16-bit bins + 256 finalists); qbuf = 64 floats of wave-private LDS.
// Returns the number of finalists, or 0xFFFFFFFF when the row must take the exact path.  One wave.
struct KbFinalArgs {          // by value: a reference to the kernel's argument struct would force it (and every pointer in it) through the stack
    const float *xp, *sq;
    unsigned k, d, magic;
};

template <int NV>
__device__ __noinline__ unsigned kb_final_row(unsigned row_dw, unsigned T, const KbFinalArgs a, size_t room_row0, size_t qglob,
                                              float sqi, unsigned qbuf_dw, int32_t *out, int lane)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kb_smem[];
    unsigned *row = (unsigned *)kb_smem + row_dw;
    float *qbuf = (float *)kb_smem + qbuf_dw;
    unsigned v[NV], bn[NV], wd[8], start;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned e = (unsigned)(i * 64 + lane);
        v[i] = e < T ? row[e] : 0xFFFFFFFFu;
    }
    // the query's 64 features in k order for the exact chains (feature k = 16 i + 4 e + g: element e of float4 [i][g])
    {
        const float *xq = a.xp + ((qglob >> 4) * 256 + (qglob & 15)) * 4;
        qbuf[lane] = xq[((lane >> 4) * 64 + 16 * (lane & 3)) * 4 + ((lane >> 2) & 3)];
    }
    wave_lds_fence();
    const KbBins B = kb_bins_of_row<NV>(v, T, lane);
    const unsigned w = (unsigned)((float)KB_MARGIN * B.scale) + 2u;          // window half width in bins
    unsigned *hist = row;
    unsigned long long *fin = (unsigned long long *)(row + 512);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned F = 0;
    if (w > 15u) {
        // the kept keys span only a few error margins: every entry is a finalist of one run
        if (T > (unsigned)KB_MAXFIN) return 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned e = (unsigned)(i * 64 + lane);
            if (e < T) fin[e] = (unsigned long long)v[i];
        }
        F = T;
    } else {
        kb_histogram<NV>(v, T, B, hist, lane, bn, wd, start);
        const unsigned d = a.d, last = (a.k - 1u) * d;
        unsigned core = 0, s = start;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned h = (wd[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
            const unsigned m = (((s + d - 1u) * a.magic) >> 18) * d;              // first multiple of d >= s
            if (h && m < s + h && m <= last) core |= 1u << t;
            s += h;
        }
        // dilate the core bins by w (< 16: the neighbours' masks suffice), then find the runs
        const unsigned prevc = lane ? (unsigned)__shfl_up((int)core, 1) : 0u;
        const unsigned nextc = lane < 63 ? (unsigned)__shfl_down((int)core, 1) : 0u;
        const unsigned long long W = (unsigned long long)prevc | ((unsigned long long)core << 16) | ((unsigned long long)nextc << 32);
        unsigned long long D = W;
        for (unsigned sh = 1; sh <= w; ++sh) D |= (W << sh) | (W >> sh);
        const unsigned flg = (unsigned)(D >> 16) & 0xFFFFu;
        const unsigned pf = lane ? (((unsigned)__shfl_up((int)flg, 1) >> 15) & 1u) : 0u;
        const unsigned rb = flg & ~((flg << 1) | pf) & 0xFFFFu;                   // bins that begin a run
        unsigned lastp = 0;
        s = start;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if ((rb >> t) & 1u) lastp = ((unsigned)(16 * lane + t + 1) << 16) | s;
            s += (wd[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
        }
        unsigned cur = kb_wave_excl_scan_max(lastp, lane);                        // the run begin in force at the lane's first bin
        s = start;
        unsigned nw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if ((rb >> t) & 1u) cur = ((unsigned)(16 * lane + t + 1) << 16) | s;
            if ((flg >> t) & 1u) nw[t >> 1] |= (0x8000u | (cur & 0x7FFFu)) << (16 * (t & 1));   // flagged: first rank of its run
            s += (wd[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ((uint2 *)hist)[4 * lane + j] = make_uint2(nw[2 * j], nw[2 * j + 1]);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool valid = (unsigned)(i * 64 + lane) < T;
            const unsigned hw = (hist[bn[i] >> 1] >> (16 * (bn[i] & 1u))) & 0xFFFFu;
            const bool take = valid && (hw & 0x8000u);
            const unsigned long long bl = __ballot(take);
            const unsigned pos = F + (unsigned)__popcll(bl & lt_mask);
            if (take && pos < (unsigned)KB_MAXFIN) fin[pos] = ((unsigned long long)(hw & 0x7FFFu) << 44) | (unsigned long long)v[i];
            F += (unsigned)__popcll(bl);
        }
    }
    wave_lds_fence();
    if (F > (unsigned)KB_MAXFIN) return 0xFFFFFFFFu;
    const float4 *xp4 = (const float4 *)a.xp;
    const float4 *q4 = (const float4 *)qbuf;
    const unsigned long long m44 = (1ull << 44) - 1ull;
    for (unsigned f0 = 0; f0 < F; f0 += 64) {
        const unsigned f = f0 + (unsigned)lane;
        const unsigned long long x = f < F ? fin[f] : 0ull;
        const unsigned idx = (unsigned)x & 0xFFFu;
        const size_t vc = room_row0 + idx;
        const float4 *pc = xp4 + (vc >> 4) * 256 + (vc & 15);
        float4 C[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) C[j] = pc[(j >> 2) * 64 + 16 * (j & 3)];
        const float sqj = a.sq[vc];
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
        const float dd = __fadd_rn(__fmaf_rn(-2.0f, z, sqi), sqj);
        if (f < F) fin[f] = (x & ~m44) | ((unsigned long long)key_of(dd) << 12) | (unsigned long long)idx;
    }
    wave_lds_fence();
    const unsigned d = a.d;
    for (unsigned f0 = 0; f0 < F; f0 += 64) {
        const unsigned f = f0 + (unsigned)lane;
        const unsigned long long x = f < F ? fin[f] : ~0ull;
        const unsigned long long xlo = x & ~m44;              // the smallest value of x's run
        // rank = first rank of the run + (finalists below x) - (finalists below the run)
        unsigned below = 0, below_run = 0;
        unsigned g = 0;
        for (; g + 4 <= F; g += 4) {
            const unsigned long long y0 = fin[g], y1 = fin[g + 1], y2 = fin[g + 2], y3 = fin[g + 3];
            below += (y0 < x ? 1u : 0u) + (y1 < x ? 1u : 0u) + (y2 < x ? 1u : 0u) + (y3 < x ? 1u : 0u);
            below_run += (y0 < xlo ? 1u : 0u) + (y1 < xlo ? 1u : 0u) + (y2 < xlo ? 1u : 0u) + (y3 < xlo ? 1u : 0u);
        }
        for (; g < F; ++g) {
            const unsigned long long y = fin[g];
            below += y < x ? 1u : 0u;
            below_run += y < xlo ? 1u : 0u;
        }
        const unsigned rank = (unsigned)(x >> 44) + below - below_run;
        const unsigned t = (rank * a.magic) >> 18;
        if (f < F && t * d == rank && t < a.k) out[t] = (int32_t)(x & 0xFFFull);
    }
    return F;
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
    float *qbuf = csq + KB_Q + 64 * (threadIdx.x >> 6);       // [KB_WAVES][64] a row's query features during its final ranking
    unsigned *smax = (unsigned *)(csq + KB_Q + 64 * KB_WAVES);   // [KB_WAVES]
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
                for (int rr = wave; rr < KB_Q; rr += KB_WAVES) {
                    const unsigned T = cnt[rr];
                    if (T > (unsigned)a.LOW) {
                        unsigned tauk;
                        const unsigned c = T <= 512u ? kb_cut_row<8>((unsigned)(rr * ROW), T, KK, lane, tauk)
                                                     : kb_cut_row<KB_NPL>((unsigned)(rr * ROW), T, KK, lane, tauk);
                        // admit iff key <= tauk, i.e. acc * m2 + ci < tauk + 1; two more units cover the float evaluation
                        if (lane == 0) {
                            cnt[rr] = c;
                            thr_f[rr] = (__fmaf_rn(csq[rr], inv_unit, KB_OFF) - (float)(tauk + 3u)) * half_unit;
                        }
                        ++st_cut;
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
        if (!need_exact) need_exact = __ballot(lane < KB_Q && cnt[lane & (KB_Q - 1)] > (unsigned)CAP) != 0ull;
        if (!need_exact) {
            bool ok = true;
            KbFinalArgs fa;
            fa.xp = a.exact.xp; fa.sq = a.sq; fa.k = (unsigned)a.k; fa.d = (unsigned)a.d; fa.magic = a.magic;
            for (int rr = wave; rr < KB_Q; rr += KB_WAVES) {
                const unsigned row = (unsigned)(rr * ROW), qb = (unsigned)(qbuf - (float *)kb_smem);
                const size_t qg = (size_t)tile * KB_Q + rr;
                int32_t *o = a.out + qg * a.k;
                const unsigned T = cnt[rr];
                const unsigned nf = T <= 512u ? kb_final_row<8>(row, T, fa, room_row0, qg, csq[rr], qb, o, lane)
                                              : kb_final_row<KB_NPL>(row, T, fa, room_row0, qg, csq[rr], qb, o, lane);
                ok = ok && nf != 0xFFFFFFFFu;
                st_fin += nf; st_ent += T;
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
    const size_t own = (size_t)KB_Q * KB_ROW * 4 + KB_Q * 12 + KB_WAVES * 64 * 4 + KB_WAVES * 4 + 16 + KB_WAVES * 64 * 4;
    return own > knn_fused_lds_bytes() ? own : knn_fused_lds_bytes();
}
