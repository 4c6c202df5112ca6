// Matrix-free dilated kNN graph in feature space (C = 64): distances, running top-(15 d + 1) selection and the final
// ranks in ONE kernel; the [N x N] distance matrix never exists (the round-1 path wrote 64 MB per block and re-read
// it).  Reference: ResGCN/gcn_lib/dense/torch_edge.py:32-59 (pairwise_distance, dense_knn_matrix: topk(-dist, k*d)) and
// :19-29 (DenseDilated: every d-th of the sorted neighbours).
//
// Included by psg_knn.hip inside its anonymous namespace, after the wave helpers (psg_wave.cuh: key_of, wave_sum_u32,
// wave_sort_keys, ...) it shares with the round-1 selection kernel.  Since round 3 this kernel is the EXACT path: the
// default is the bf16-prefilter kernel (psg_knn_bf.cuh), which calls knn_exact_block below for the tiles it cannot decide.
//
// Work split.  A workgroup of 16 waves owns 16 QUERY points of one room and streams all N candidates of that room past
// them; wave w takes candidate pairs-of-tiles w, w + 16, ... (32 candidates per wave and step, 512 per workgroup and
// step).  Four waves share a SIMD: while one waits for its operands, an LDS atomic or a barrier, the others keep the
// matrix pipe busy (the first version had one wave per SIMD and spent 2.7x the MFMA time in the stream, everything else
// latency-exposed on top).  Distance tiles come from v_mfma_f32_16x16x4_f32 with A = 16 candidates x K and B = K x 16
// queries, so lane l holds, for query l & 15, the four candidates 4 (l >> 4) + r: the query sits on the lane and
// everything that follows is per-lane work.  The dot product is the ascending-k fmaf chain from 0 (16 chained MFMAs,
// k = 4 s + (l >> 4) inside step s; tools/mfma_chain_probe.hip: bit-identical to the host's fmaf chain on 409 600 dots),
// the distance is (|x_i|^2 + (-2 x_i.x_j)) + |x_j|^2 with torch's rounding: equal features give bit-equal distances.
// Operands are read straight from L2 in MFMA operand order ("xp": [point / 16][4][64 lanes] float4, element e of
// lane l in quarter i = feature 4 (4 i + e) + (l >> 4) of point (l & 15)), one coalesced 1-KiB load per quarter; the
// producer of the features (edge_max_fwd_kernel) writes that copy, knn_prep_kernel does it for the stand-alone entry
// (psg_knn_ops.cuh, psg_knn.hip).
//
// Selection.  Per query row a buffer of CAP = 1024 composite keys (distance key << 12 | candidate index: ascending
// distance, lowest index first, a strict total order) lives in LDS (rows 1026 keys apart - see KF_ROW -, 128 KB per
// workgroup, one workgroup per CU) with a threshold tau.  In the stream a pair is admitted by one float compare, not
// (distance > tau's distance): a superset of "key <= tau" (equal distances with a higher index and NaNs get in too; they rank
// behind the keys wanted and are dropped by the next cut); slots come from one LDS atomic per lane and step.  On a
// data-independent schedule the workgroup PRUNES: one wave per row finds (regula falsi on the distance value, counting
// with DPP wave sums) a new tau with KK <= #(keys <= tau) <= KK + TOL and compacts the row in place.  tau is never below
// the KK-th smallest key seen so far, so the buffer always contains the true top-KK of what has been seen: exact for any
// input order.  Everything is admitted until the first prune (after 1024 candidates); after it a row receives
// ~KK ln(N / 1024) more keys, so typical rows are pruned once or twice.  An allocation that does not fit is not written;
// the counter then exceeds CAP, the next synchronisation point sees it and the workgroup redoes its stream on the exact
// per-step path: exact composite admission, a barrier per step, and on overflow the step is rolled back, every row is cut
// to exactly KK and the step's candidates are committed one accumulator register at a time (<= 64 appends per row
// between checks, KK + 64 <= CAP): slow, but it terminates and is exact; ties are ordered by index, so an exact cut
// always reaches KK.
//
// Final ranks.  Only ranks 0, d, .., 15 d are wanted, so the row is not sorted (a 512-key bitonic sort through
// ds_bpermute cost 34k cycles per row): the row's keys (up to 1024, in registers: the row's LDS is scratch by then) are
// binned by distance (1024 linear bins, a monotone map, so bin order = key order), an exclusive scan gives every bin its
// first rank, the bins that contain a wanted rank are flagged, their few members ("finalists") are collected and each
// finalist counts the finalists of its own bin below it: first rank of the bin + that count is its exact rank.  More
// than 256 finalists (hundreds of equal distances) fall back to a cut to 512 keys + the bitonic sort.
#pragma once

#ifndef KF_WAVES_CFG
#define KF_WAVES_CFG 16
#endif
constexpr int KF_WAVES = KF_WAVES_CFG;
constexpr int KF_Q = 16;        // query rows per workgroup (one 16-column MFMA tile)
constexpr int KF_STEP = 32;     // candidates per wave and step: two 16-row MFMA tiles (two independent accumulators)
#ifndef KF_CAP_CFG
#define KF_CAP_CFG 1024
#endif
constexpr int KF_CAP = KF_CAP_CFG;    // composite keys per row buffer
constexpr int KF_NPL = KF_CAP / 64;
// Row buffers are KF_ROW keys apart, not KF_CAP: with a stride of 8 KiB the 16 rows start in the same LDS bank, and while
// everything is admitted (all rows at the same fill) every append instruction was a 16-way bank conflict: the first two
// steps took 12k cycles each instead of 7.6k.  16 bytes of skew per row spread the rows over all banks.
constexpr int KF_ROW = KF_CAP + 2;

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Diagnostic build only (make EXTRA=-DPSG_KF_STAMP, tools/knn_stamp.py): per-phase cycle sums of wave 0 of every
// workgroup, written to a buffer no other code reads.  Phases: 0 stream (operands + MFMA + keys + slot allocation),
// 1 wait at the step barrier, 2 appends, 3 prune events, 4 final ranks, 5 number of prune events, 6 roll-backs.
#ifdef PSG_KF_STAMP
__device__ unsigned long long g_kf_stamps[16];
#define KF_T(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define KF_ACC(i, t1, t0) kf_sum[i] += (t1) - (t0)
#else
#define KF_T(var)
#define KF_ACC(i, t1, t0)
#endif

// Second diagnostic build (EXTRA=-DPSG_KF_TL, tools/knn_timeline.py): no waits inside the steps; thread 0 adds the time
// since kernel start at every workgroup barrier exit, so differences are the real lengths of the workgroup-wide phases.
#ifdef PSG_KF_TL
__device__ unsigned long long g_kf_tl[32];
#define KF_TL(idx) do { if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    atomicAdd(&g_kf_tl[(idx)], t_ - tl0); atomicAdd(&g_kf_tl[16 + (idx)], 1ull); } } while (0)
#else
#define KF_TL(idx)
#endif

struct KnnFusedArgs {
    const float *xp;   // [rows / 16][4][64] float4, operand order (see above); rows = B * N
    const float *sq;   // [rows] squared norms in torch.sum's order
    int32_t *out;      // [rows][k]
    int N;             // points per room, multiple of 16
    int k, d;
    int KK;            // (k - 1) d + 1 keys decide the output (<= 448)
    unsigned magic;    // ceil(2^18 / d): x / d = (x * magic) >> 18 for x < 2^18 / d
    int LOW, TOL;      // a prune event cuts the rows above LOW to [KK, KK + TOL]
    int xcd_order;     // 1: XCD-aware block order (default); 0: identity (PSG_KNN_XCD_ORDER=0, A/B and diagnosis)
    unsigned rsel;     // 0: rows are cut to KK on the growth schedule.  > 0 (round 4): ONE cut, after the first CAP candidates, to the
                       // rsel smallest keys - a sampled threshold as in psg_knn_bf.cuh: wave w streams the pairs w S .. w S + S - 1, so
                       // those CAP candidates are spread over the room; a row that ends with fewer than KK keys takes the per-step path
};

// inclusive prefix sum over the 64 lanes (row scans by DPP row_shr, then the row totals by row_bcast)
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v)
{
    v += dpp_get<0x111, 0xF>(v);   // row_shr:1 (lanes shifted in from outside the row read 0)
    v += dpp_get<0x112, 0xF>(v);   // row_shr:2
    v += dpp_get<0x114, 0xF>(v);   // row_shr:4
    v += dpp_get<0x118, 0xF>(v);   // row_shr:8: inclusive scan inside each row of 16 lanes
    v += dpp_get<0x142, 0xA>(v);   // row_bcast15: rows 1, 3 += total of the row before
    v += dpp_get<0x143, 0xC>(v);   // row_bcast31: rows 2, 3 += total of rows 0..1
    return v;
}

// Cut row `ent` (T composite keys, T <= 64 * NPL) to the C smallest with KK <= C <= KK + tol (tol = 0: exactly KK),
// compacted in place; returns C and the threshold (every kept key <= thr, every dropped key > thr).  One wave.
// Regula falsi on the distance value: 4 to 6 counting rounds.
template <int NPL>
__device__ __forceinline__ unsigned prune_row(unsigned long long *ent, unsigned T, unsigned KK, unsigned tol, int lane,
                                              unsigned long long &thr_out)
{
    unsigned long long v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const unsigned e = (unsigned)(i * 64 + lane);
        v[i] = e < T ? ent[e] : ~0ull;
    }
    // bracket from the 32-bit distance keys (DPP reductions; a 64-bit min / max through ds_bpermute cost ~3k cycles)
    unsigned kmn = 0xFFFFFFFFu, kmx = 0u;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const unsigned kq = (unsigned)(v[i] >> 12);
        kmn = kq < kmn ? kq : kmn;
        kmx = (v[i] != ~0ull && kq > kmx) ? kq : kmx;
    }
    kmn = ~wave_max_u32(~kmn);
    kmx = wave_max_u32(kmx);
    // invariant: #(v <= lo) = c_lo < KK <= #(v <= hi) = c_hi
    unsigned long long lo = ((unsigned long long)kmn << 12) - 1ull, hi = ((unsigned long long)kmx << 12) | 0xFFFull;
    unsigned c_hi = T, c_lo = 0, it = 0;
    while (c_hi > KK + tol && hi - lo > 1ull) {
        // probe in DISTANCE space: interpolate the wanted count between the bracket's counts (regula falsi; every third
        // probe the plain midpoint, so one-sided convergence cannot stall), all indices admitted at that distance;
        // integer midpoint of the composite keys when the distance probe does not split the bracket (ties)
        const float dl = dist_of((unsigned)(lo >> 12)), dh = dist_of((unsigned)(hi >> 12));
        float frac = ((float)(KK + (tol >> 1)) + 0.5f - (float)c_lo) / (float)(c_hi - c_lo);
        frac = (it % 3u == 2u) ? 0.5f : fminf(fmaxf(frac, 0.02f), 0.98f);
        ++it;
        const float dm = dl + (dh - dl) * frac;
        unsigned long long mid = ((unsigned long long)key_of(dm) << 12) | 0xFFFull;
        if (!(mid > lo && mid < hi)) mid = lo + ((hi - lo) >> 1);
        unsigned c = 0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) c += v[i] <= mid ? 1u : 0u;
        c = wave_sum_u32(c);
        if (c < KK) { lo = mid; c_lo = c; }
        else { hi = mid; c_hi = c; }
    }
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned base = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const bool keep = v[i] <= hi;
        const unsigned long long b = __ballot(keep);
        if (keep) ent[base + __popcll(b & lt_mask)] = v[i];
        base += (unsigned)__popcll(b);
    }
    thr_out = hi;
    return c_hi;
}

__device__ __forceinline__ unsigned kf_bin(unsigned long long comp, float dmin, float scale)
{
    const float t = (dist_of((unsigned)(comp >> 12)) - dmin) * scale;
    return (unsigned)fminf(fmaxf(t, 0.0f), 1023.0f);      // (NaN -> 0: fmaxf returns the other operand)
}

// Exact ranks of a few of a row's keys without sorting it.  The row's keys are in registers (v, NV per lane, ~0 = none),
// so the row's own LDS memory is scratch: `hist` = 512 words (1024 16-bit bins), `fin` = 256 finalists.  Keys are binned
// by distance (1024 linear bins between the row's extremes: a monotone map, so bin order = key order), an exclusive scan
// gives every bin its first rank, wanted(first_rank, size) flags the bins that hold a rank the caller wants, the members of
// flagged bins ("finalists") are collected and each counts the finalists of its own bin below it: first rank of the
// bin + that count is its exact rank, handed to emit(rank, key).  Returns false (nothing emitted) when there are more
// than 256 finalists (hundreds of equal or crowded distances): the caller falls back to a slower exact method.  One wave.
template <int NV, class Wanted, class Emit>
__device__ __forceinline__ bool hist_ranks(const unsigned long long (&v)[NV], unsigned *hist, unsigned long long *fin,
                                           unsigned *scratch_cnt, int lane, Wanted wanted, Emit emit)
{
    unsigned kmn = 0xFFFFFFFFu, kmx = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned kq = (unsigned)(v[i] >> 12);
        kmn = kq < kmn ? kq : kmn;
        kmx = (v[i] != ~0ull && kq > kmx) ? kq : kmx;
    }
    kmn = ~wave_max_u32(~kmn);
    kmx = wave_max_u32(kmx);
    const float dmin = dist_of(kmn), dmax = dist_of(kmx);
    const float scale = dmax > dmin ? 1023.0f / (dmax - dmin) : 0.0f;
    ((uint4 *)hist)[2 * lane] = make_uint4(0u, 0u, 0u, 0u);
    ((uint4 *)hist)[2 * lane + 1] = make_uint4(0u, 0u, 0u, 0u);
    if (lane == 0) *scratch_cnt = 0u;
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned b = kf_bin(v[i], dmin, scale);
        if (v[i] != ~0ull) atomicAdd(&hist[b >> 1], 1u << (16 * (b & 1u)));
    }
    wave_lds_fence();
    // lane owns bins 16 lane .. 16 lane + 15 (words 8 lane .. 8 lane + 7)
    uint4 w0 = ((const uint4 *)hist)[2 * lane], w1 = ((const uint4 *)hist)[2 * lane + 1];
    unsigned w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    unsigned tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += (w[j] & 0xFFFFu) + (w[j] >> 16);
    unsigned start = wave_incl_scan_u32(tot) - tot;      // first rank of the lane's first bin
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned nw = 0;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const unsigned h = hlf ? (w[j] >> 16) : (w[j] & 0xFFFFu);
            unsigned hw = h;
            if (h && wanted(start, h)) hw = 0x8000u | start;                      // flagged: holds the bin's first rank
            nw |= hw << (16 * hlf);
            start += h;
        }
        w[j] = nw;
    }
    ((uint4 *)hist)[2 * lane] = make_uint4(w[0], w[1], w[2], w[3]);
    ((uint4 *)hist)[2 * lane + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (v[i] != ~0ull) {
            const unsigned b = kf_bin(v[i], dmin, scale);            // (recomputed: 16 registers less than keeping it)
            const unsigned hw = (hist[b >> 1] >> (16 * (b & 1u))) & 0xFFFFu;
            if (hw & 0x8000u) {
                const unsigned pos = atomicAdd(scratch_cnt, 1u);
                if (pos < 256u) fin[pos] = ((unsigned long long)(hw & 0x7FFFu) << 44) | v[i];
            }
        }
    }
    wave_lds_fence();
    const unsigned F = *scratch_cnt;
    if (F > 256u) return false;
    const unsigned long long m44 = (1ull << 44) - 1ull;
    for (unsigned f = lane; f < F; f += 64) {
        const unsigned long long x = fin[f];
        const unsigned st = (unsigned)(x >> 44);
        const unsigned long long xc = x & m44;
        unsigned rank = st;
        for (unsigned g = 0; g < F; ++g) {
            const unsigned long long y = fin[g];
            rank += ((unsigned)(y >> 44) == st && (y & m44) < xc) ? 1u : 0u;
        }
        emit(rank, xc);
    }
    return true;
}

// Ranks 0, d, .., (k-1) d of a row's T >= KK composite keys (see the header); T <= 64 NV.  One wave; `row` has KF_CAP
// slots.
template <int NV>
__device__ __forceinline__ void final_ranks(unsigned long long *row, unsigned T, const KnnFusedArgs &a, unsigned *scratch_cnt,
                                            int32_t *out, int lane)
{
    const unsigned KK = (unsigned)a.KK;
    unsigned long long v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned e = (unsigned)(i * 64 + lane);
        v[i] = e < T ? row[e] : ~0ull;
    }
    wave_lds_fence();
    const unsigned d = (unsigned)a.d, last = (unsigned)(a.k - 1) * d;
    const bool ok = hist_ranks<NV>(v, (unsigned *)row, row + 256, scratch_cnt, lane,
        [&](unsigned start, unsigned h) {
            const unsigned m = (((start + d - 1u) * a.magic) >> 18) * d;              // first multiple of d >= start
            return m < start + h && m <= last;
        },
        [&](unsigned rank, unsigned long long xc) {
            const unsigned t = (rank * a.magic) >> 18;
            if (t * d == rank && t < (unsigned)a.k) out[t] = (int32_t)(xc & 0xFFFull);
        });
    if (ok) return;
    // hundreds of equal distances: sort the (at most 512 smallest) keys; padding sorts last
#pragma unroll
    for (int i = 0; i < NV; ++i) row[i * 64 + lane] = v[i];
    wave_lds_fence();
    if (NV > 8 && T > 512u) {
        unsigned long long thr;
        const unsigned room = 512u - KK;
        T = prune_row<NV>(row, T, KK, room < 64u ? room : 64u, lane, thr);
        wave_lds_fence();
        for (unsigned t = T + lane; t < 512u; t += 64) row[t] = ~0ull;
        wave_lds_fence();
    }
    wave_sort_keys<8>(row, lane);
    wave_lds_fence();
    if (lane < a.k) out[lane] = (int32_t)(row[(size_t)lane * d] & 0xFFFull);
}

// The whole exact path for the 16 queries [16 qblk, 16 qblk + 16) by one workgroup of KF_WAVES waves; kf_smem = the
// workgroup's dynamic LDS (knn_fused_lds_bytes()).  Called by knn_fused_kernel (one block per workgroup) and, as the exact
// fallback for a 32-query tile, by the bf16-prefilter kernel (psg_knn_bf.cuh).  Ends with every output of the block stored;
// the caller synchronises the workgroup before it reuses the LDS.
__device__ __forceinline__ void knn_exact_block(const KnnFusedArgs &a, const int qblk, unsigned char *kf_smem)
{
    constexpr int CAP = KF_CAP, ROW = KF_ROW;
    constexpr int PER_STEP = KF_WAVES * KF_STEP;                             // candidates per workgroup and step
    unsigned long long *ent = (unsigned long long *)kf_smem;                 // [KF_Q][ROW], CAP used
    unsigned *tau_k = (unsigned *)(ent + KF_Q * ROW);                        // [KF_Q] threshold: distance key ..
    unsigned *tau_i = tau_k + KF_Q;                                          // [KF_Q] .. and index (ties)
    unsigned *cnt = tau_i + KF_Q;                                            // [KF_Q]
    unsigned *flag = cnt + KF_Q;                                             // [3][2]: overflow, prune (by step % 3)
    unsigned *fcnt = flag + 8;                                               // [KF_Q] finalists per row

    // (readfirstlane: the wave number is uniform, so tile numbers and operand addresses are scalar work)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, g = lane >> 4;
    KF_T(t_begin);
    const int tiles_per_room = a.N >> 4;
    const int room = qblk / tiles_per_room;
    const size_t room_row0 = (size_t)room * a.N;
    const size_t tile0 = (size_t)room * tiles_per_room;                      // first operand tile of the room
    const size_t qrow = (size_t)qblk * KF_Q + q;
    const unsigned KK = (unsigned)a.KK;

    if (tid < KF_Q) { tau_k[tid] = 0xFFFFFFFFu; tau_i[tid] = 0xFFFu; cnt[tid] = 0u; }      // everything is admitted
    if (tid < 6) flag[tid] = 0u;

    const float4 *xp4 = (const float4 *)a.xp;
    const float4 *sq4 = (const float4 *)(a.sq + room_row0);
    float4 bq[4];
    auto load_queries = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) bq[i] = xp4[((size_t)qblk * 4 + i) * 64 + lane];
    };
    load_queries();
    const float sqi = a.sq[qrow];

    const int n_pairs = (a.N + KF_STEP - 1) / KF_STEP;                       // candidate pairs-of-tiles in the room
    const int n_steps = (n_pairs + KF_WAVES - 1) / KF_WAVES;
    const int last_tile = tiles_per_room - 1;
    // candidate pair-of-tiles of this wave at a step: round robin, or (sampled threshold) a contiguous range per wave
    const bool sampled = a.rsel != 0u;
    auto pair_of = [&](int step) { return sampled ? wave * n_steps + step : step * KF_WAVES + wave; };
#ifdef PSG_KF_TL
    unsigned long long tl0 = 0;
    if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl0) :: "memory");
#endif
    __syncthreads();
    KF_TL(0);

#ifdef PSG_KF_STAMP
    unsigned long long kf_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // operands (two candidate tiles in MFMA operand order and their squared norms), 32 MFMAs and the 8 distances of the
    // lane for one step.  (Loading the next step's operands early, into the registers a chain has just consumed, was
    // tried twice and lost 6 %: the step is bound by the sum of matrix, vector and operand-path time, not by latency.)
    auto step_compute = [&](int step, float (&dd)[8], int &c0) {
        KF_T(t_a);
        const int pair = pair_of(step);
        const int t0 = min(2 * pair, last_tile), t1 = min(2 * pair + 1, last_tile);   // clamped: masked out below
        float4 a0[4], a1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[i] = xp4[((tile0 + t0) * 4 + i) * 64 + lane];
            a1[i] = xp4[((tile0 + t1) * 4 + i) * 64 + lane];
        }
        const float4 s0 = sq4[t0 * 4 + g], s1 = sq4[t1 * 4 + g];
#ifdef PSG_KF_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KF_T(t_a1);
        KF_ACC(8, t_a1, t_a);          // operands landed
#endif
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].x, bq[i].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].x, bq[i].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].y, bq[i].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].y, bq[i].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].z, bq[i].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].z, bq[i].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].w, bq[i].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].w, bq[i].w, acc1, 0, 0, 0);
        }
        const float sj[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#ifdef PSG_KF_STAMP
        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(acc0[3]), "+v"(acc1[3]));   // waits for the MFMA chains
        KF_T(t_a2);
        KF_ACC(9, t_a2, t_a1);         // MFMA chains done
#endif

        // ---- distances of the lane's 8 (query, candidate) pairs: candidates c0 + r (tile 0) and c0 + 16 + r (tile 1)
        c0 = pair * KF_STEP + 4 * g;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float z = r < 4 ? acc0[r] : acc1[r - 4];
            // (|x_i|^2 + (-2 x_i.x_j)) + |x_j|^2, torch_edge.py:41-43 (-2 z is exact, so the fma rounds once like the add)
            dd[r] = __fadd_rn(__fmaf_rn(-2.0f, z, sqi), sj[r]);
        }
#ifdef PSG_KF_STAMP
        asm volatile("" :: "v"(dd[7]));
        KF_T(t_a3);
        KF_ACC(10, t_a3, t_a2);        // distances
        KF_ACC(0, t_a3, t_a);
#endif
    };
    // exact admission: keys and the mask of the pairs with (key, index) <= tau (the per-step path)
    auto exact_filter = [&](int step, const float (&dd)[8], int c0, unsigned (&key)[8], unsigned &mask) {
        const unsigned tk = tau_k[q], ti = tau_i[q];
        mask = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            key[r] = key_of(dd[r]);
            const unsigned c = (unsigned)(c0 + (r < 4 ? r : 12 + r));
            mask |= ((key[r] < tk || (key[r] == tk && c <= ti)) ? 1u : 0u) << r;
        }
        if (pair_of(step) * KF_STEP + KF_STEP > a.N) {   // ragged tail (wave-uniform): candidates beyond N never pass
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (c0 + (r < 4 ? r : 12 + r) >= a.N) mask &= ~(1u << r);
        }
    };

    // ---- FAST PATH: no barrier per step.  Slots are taken with the same atomic; an allocation that does not fit is simply
    // not written (the row's counter then exceeds CAP, which is detected at the next synchronisation point and sends the
    // whole workgroup through the exact per-step path below: adversarial candidate orders only).  Rows are pruned at a
    // data-independent schedule (below); between those points the 16 waves run free, so the four waves of a SIMD overlap
    // operand loads, MFMAs and key / append work by themselves.
    bool need_safe = false;
    {
        // prune schedule, in candidates seen: first when the buffer is full (CAP, everything was admitted); a row cut to
        // c <= KK + TOL keys at n_i admits about c / n_i of what follows (tau is fixed until the next prune), so the next
        // prune comes when c * n / n_i reaches 0.9 CAP: n_{i+1} = 0.9 CAP / (KK + TOL) * n_i, in whole steps
        const float grow = 0.9f * (float)CAP / (float)(KK + (unsigned)a.TOL);
        unsigned next_prune = (unsigned)CAP;
#ifdef PSG_KF_TL
        int n_tl = 0;
#endif
        for (int step = 0; step < n_steps; ++step) {
            float dd[8];
            int c0;
            step_compute(step, dd, c0);
            KF_T(t_s1);
            // Admission by ONE float compare per pair: not (distance > tau's distance).  A superset of the exact rule
            // (it lets equal distances with a higher index and NaNs in; an open row's tau reads as NaN and admits
            // everything): extra keys rank behind the ones wanted and are dropped by the next prune or the final
            // ranking; rows they would overflow go through the exact per-step path like any other overflow.  The key and
            // the composite are only built for the few pairs that pass.
            const float tf = dist_of(tau_k[q]);
            if (pair_of(step) * KF_STEP + KF_STEP <= a.N) {
                unsigned n = 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) n += !(dd[r] > tf) ? 1u : 0u;
                if (n) {
                    const unsigned base = atomicAdd(&cnt[q], n);
                    if (base + n <= (unsigned)CAP) {
                        unsigned long long *dst = ent + q * ROW + base;
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            if (!(dd[r] > tf))
                                *dst++ = ((unsigned long long)key_of(dd[r]) << 12) | (unsigned)(c0 + (r < 4 ? r : 12 + r));
                    }
                }
            } else {
                // ragged tail (wave-uniform, N not a multiple of 32): the exact rule, candidates beyond N masked out
                unsigned key[8], mask;
                exact_filter(step, dd, c0, key, mask);
                const unsigned n = (unsigned)__popc(mask);
                if (n) {
                    const unsigned base = atomicAdd(&cnt[q], n);
                    if (base + n <= (unsigned)CAP) {
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            if ((mask >> r) & 1u)
                                ent[q * ROW + base + __popc(mask & ((1u << r) - 1u))] =
                                    ((unsigned long long)key[r] << 12) | (unsigned)(c0 + (r < 4 ? r : 12 + r));
                    }
                }
            }
            KF_T(t_s2);
            KF_ACC(2, t_s2, t_s1);
            const unsigned seen = (unsigned)(step + 1) * (unsigned)PER_STEP;
            const bool sync_now = seen >= next_prune && step + 1 < n_steps;
            if (sync_now) {
                const unsigned nn = (unsigned)(grow * (float)seen) / (unsigned)PER_STEP * (unsigned)PER_STEP;
                next_prune = sampled ? 0xFFFFFFFFu : (nn > seen ? nn : seen + (unsigned)PER_STEP);
                __syncthreads();
                KF_TL(n_tl < 3 ? 1 + 2 * n_tl : 5);
                // (every wave reads the same counters, and no wave rewrites one before all have passed this test: a row is
                // only pruned by waves that saw no overflow, and an overflowing counter stays above CAP)
                if (__ballot(lane < KF_Q && cnt[lane & (KF_Q - 1)] > (unsigned)CAP) != 0ull) { need_safe = true; break; }
                for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
                    const unsigned T = cnt[qq];
                    if (T > (unsigned)a.LOW) {
                        unsigned long long thr;
                        const unsigned c = prune_row<KF_NPL>(ent + qq * ROW, T, sampled ? a.rsel : KK, (unsigned)a.TOL, lane, thr);
                        if (lane == 0) { cnt[qq] = c; tau_k[qq] = (unsigned)(thr >> 12); tau_i[qq] = (unsigned)thr & 0xFFFu; }
                    }
                }
                __syncthreads();
#ifdef PSG_KF_STAMP
                kf_sum[5] += 1;
#endif
                KF_T(t_s3);
                KF_ACC(3, t_s3, t_s2);
                KF_TL(n_tl < 3 ? 2 + 2 * n_tl : 6);
#ifdef PSG_KF_TL
                ++n_tl;
#endif
            }
        }
        __syncthreads();
        KF_TL(8);
        // (sampled threshold: the row holds every candidate at or below its threshold, so KK of them prove it complete; a row
        // the sample left short - candidates sorted by distance, one row in 10^7 otherwise - starts over on the per-step path)
        if (!need_safe) need_safe = __ballot(lane < KF_Q && (cnt[lane & (KF_Q - 1)] > (unsigned)CAP || (sampled && cnt[lane & (KF_Q - 1)] < KK))) != 0ull;
    }
    if (need_safe) {
        // ---- EXACT PER-STEP PATH (rare): start over with a barrier per step, roll-back on overflow
        __syncthreads();
        if (tid < KF_Q) { tau_k[tid] = 0xFFFFFFFFu; tau_i[tid] = 0xFFFu; cnt[tid] = 0u; }
        if (tid < 6) flag[tid] = 0u;
        __syncthreads();
#ifdef PSG_KF_STAMP
        kf_sum[6] += 1;
#endif
    for (int step = 0; step < n_steps; ++step) {
        float dd[8];
        unsigned key[8], mask;
        int c0;
        step_compute(step, dd, c0);
        exact_filter(step, dd, c0, key, mask);
        const unsigned n = (unsigned)__popc(mask);
        unsigned base = 0;
        if (n) base = atomicAdd(&cnt[q], n);
        // room the NEXT step may need: about PER_STEP * KK / (candidates seen) keys per row, doubled, at most PER_STEP
        const unsigned need = 2u * KK / (unsigned)(step + 1) + 32u;
        const unsigned high = (unsigned)CAP - (need < (unsigned)PER_STEP ? need : (unsigned)PER_STEP);
        const int ph = step % 3;
        if (__ballot(n && base + n > (unsigned)CAP) && lane == 0) flag[2 * ph] = 1u;
        if (__ballot(n && base + n > high) && lane == 0) flag[2 * ph + 1] = 1u;
        KF_T(t_b);
        // (not __syncthreads(): that would also wait for the operand loads in flight; only LDS traffic must have landed)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        KF_T(t_c);
        KF_ACC(1, t_c, t_b);
        const unsigned f_ovf = flag[2 * ph], f_prune = flag[2 * ph + 1];
        if (tid == 0) { flag[2 * ((step + 2) % 3)] = 0u; flag[2 * ((step + 2) % 3) + 1] = 0u; }

        if (!f_ovf) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if ((mask >> r) & 1u)
                    ent[q * ROW + base + __popc(mask & ((1u << r) - 1u))] =
                        ((unsigned long long)key[r] << 12) | (unsigned)(c0 + (r < 4 ? r : 12 + r));
            KF_T(t_d);
            KF_ACC(2, t_d, t_c);
            if (f_prune && step + 1 < n_steps) {
                __syncthreads();
                for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
                    const unsigned T = cnt[qq];
                    if (T > (unsigned)a.LOW) {
                        unsigned long long thr;
                        const unsigned c = prune_row<KF_NPL>(ent + qq * ROW, T, KK, (unsigned)a.TOL, lane, thr);
                        if (lane == 0) { cnt[qq] = c; tau_k[qq] = (unsigned)(thr >> 12); tau_i[qq] = (unsigned)thr & 0xFFFu; }
                    }
                }
                __syncthreads();
#ifdef PSG_KF_STAMP
                KF_T(t_e);
                KF_ACC(3, t_e, t_d);
                kf_sum[5] += 1;
#endif
            }
        } else {
            // ---- overflow (adversarial orders): roll the step back, cut every row to exactly KK, commit one register
            // at a time (<= 64 appends per row between checks)
            if (n) atomicSub(&cnt[q], n);
            __syncthreads();
            for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
                const unsigned T = cnt[qq];
                if (T > KK) {
                    unsigned long long thr;
                    const unsigned c = prune_row<KF_NPL>(ent + qq * ROW, T, KK, 0u, lane, thr);
                    if (lane == 0) { cnt[qq] = c; tau_k[qq] = (unsigned)(thr >> 12); tau_i[qq] = (unsigned)thr & 0xFFFu; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const unsigned c = (unsigned)(c0 + (r < 4 ? r : 12 + r));
                const bool pass = (int)c < a.N && (key[r] < tau_k[q] || (key[r] == tau_k[q] && c <= tau_i[q]));
                if (pass) ent[q * ROW + atomicAdd(&cnt[q], 1u)] = ((unsigned long long)key[r] << 12) | c;
                __syncthreads();
                for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
                    const unsigned T = cnt[qq];
                    if (T + 64u > (unsigned)CAP) {
                        unsigned long long thr;
                        const unsigned c = prune_row<KF_NPL>(ent + qq * ROW, T, KK, 0u, lane, thr);
                        if (lane == 0) { cnt[qq] = c; tau_k[qq] = (unsigned)(thr >> 12); tau_i[qq] = (unsigned)thr & 0xFFFu; }
                    }
                }
                __syncthreads();
            }
        }
    }
    }   // need_safe
    __syncthreads();
    KF_T(t_f);
    for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
        unsigned long long *row = ent + qq * ROW;
        int32_t *o = a.out + ((size_t)qblk * KF_Q + qq) * a.k;
        const unsigned T = cnt[qq];
        if (T <= 512u) final_ranks<8>(row, T, a, fcnt + qq, o, lane);
        else final_ranks<KF_NPL>(row, T, a, fcnt + qq, o, lane);
    }
    KF_TL(9);
#ifdef PSG_KF_TL
    __syncthreads();
    KF_TL(10);
#endif
#ifdef PSG_KF_STAMP
    KF_T(t_g);
    KF_ACC(4, t_g, t_f);
    KF_ACC(13, t_g, t_begin);
    if (tid == 0)
        for (int i = 0; i < 14; ++i) atomicAdd(&g_kf_stamps[i], kf_sum[i]);
#endif
}

// Registers: 5 waves per SIMD asked of the compiler although a workgroup only brings 4 - the kernel then takes 96 VGPRs
// (4 spilled, outside the stream loop) instead of 122, which leaves a quarter of every SIMD's register file and 31 KB of LDS
// to workgroups of OTHER kernels.  At 122 VGPRs a resident kNN workgroup filled the register file and nothing else could
// start on its CU: the small GEMMs and edge passes of the other launches in flight waited for the kNN kernel's tail.
// Same kernel time (4.47 ms per 4-room iteration), 12.02 -> 12.36 rooms/s; at 6 waves (80 VGPRs, 16 spills) the kernel
// itself slows down by 9 %.
#ifndef KF_MIN_WAVES_EU
#define KF_MIN_WAVES_EU 5
#endif
__global__ __launch_bounds__(KF_WAVES * 64, KF_MIN_WAVES_EU) void knn_fused_kernel(KnnFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kf_smem[];
    // XCD-aware block order: blocks b, b + 8, .. share an XCD (round-robin dispatch), so XCD x takes the contiguous query
    // blocks [x G / 8, (x + 1) G / 8): a room's operand copy is fetched into one or two L2s instead of all eight (speed
    // only; round 2 measured 35.5 MB of HBM traffic per 4-room launch against 5.3 MB algorithmic with the identity order)
    const unsigned G = gridDim.x, b = blockIdx.x;
    knn_exact_block(a, (int)((G & 7u) == 0u && a.xcd_order ? (b & 7u) * (G >> 3) + (b >> 3) : b), kf_smem);
}

// the same as a call (the prefilter kernel's cold path: inlined twice it doubled that kernel and spilled its stream loop)
__device__ __noinline__ void knn_exact_block_cold(const KnnFusedArgs a, const int qblk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kf_smem[];
    knn_exact_block(a, qblk, kf_smem);
}

inline size_t knn_fused_lds_bytes() { return (size_t)KF_Q * KF_ROW * 8 + KF_Q * 8 + KF_Q * 4 + (8 + KF_Q) * 4; }
