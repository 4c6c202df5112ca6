// Matrix-free dilated kNN graph in feature space (C = 64): distances, running top-(15 d + 1) selection and the final
// ordering in ONE kernel; the [N x N] distance matrix never exists (the round-1 path wrote 64 MB per block and re-read
// it).  Reference: ResGCN/gcn_lib/dense/torch_edge.py:32-59 (pairwise_distance, dense_knn_matrix: topk(-dist, k*d)) and
// :19-29 (DenseDilated: every d-th of the sorted neighbours).
//
// Included by psg_resgcn.hip inside its anonymous namespace, after the wave helpers (key_of, wave_sum_u32,
// wave_sort_keys, ...) it shares with the round-1 selection kernel.
//
// Work split.  A workgroup (4 waves) owns 16 QUERY points of one room and streams all N candidates of that room past
// them; wave w takes candidate pairs-of-tiles w, w + 4, ... (32 candidates per step).  Distance tiles come from
// v_mfma_f32_16x16x4_f32 with A = 16 candidates x K and B = K x 16 queries, so lane l holds, for query l & 15, the four
// candidates 4 (l >> 4) + r: the query sits on the lane and everything that follows is per-lane work.  The dot product is
// the ascending-k fmaf chain from 0 (16 chained MFMAs, k = 4 s + (l >> 4) inside step s: the same chain as a CPU sgemm,
// SURVEY.md 8a'), the distance is (|x_i|^2 + (-2 x_i.x_j)) + |x_j|^2 with torch's rounding; equal features give
// bit-equal distances, as with the round-1 GEMM.
// Operands are read straight from L2 in MFMA operand order ("xp": [point / 16][4][64 lanes] float4, element e of
// lane l in quarter i = feature 4 (4 i + e) + (l >> 4) of point (l & 15)), one coalesced 1-KiB load per quarter; the
// producer of the features (edge_max_fwd_kernel) writes that copy, knn_prep_kernel does it for the stand-alone entry
// point.  The next step's operands are in flight under this step's MFMAs.
//
// Selection.  Per query row a buffer of CAP composite keys (distance key << 12 | candidate index: ascending distance,
// lowest index first, a strict total order) lives in LDS together with a threshold tau.  A candidate is appended iff
// its key <= tau; slots come from one LDS atomic per lane and step.  When a row runs full the workgroup PRUNES it: a
// wave finds (bisection on the distance value, counting with DPP wave sums) a new tau with KK <= #(keys <= tau) <=
// KK + TOL and compacts the row in place.  tau is always the KK-th smallest of the candidates seen so far or larger, so
// the buffer always contains the true top-KK of what has been seen: exact, for any input order.  Expected appends per
// row ~ KK (1 + ln(N / KK)) for random order.  If an append would overflow CAP the step is rolled back, every row is
// pruned exactly to KK and the step's candidates are committed one accumulator register at a time (at most 16 appends
// per row between checks, KK + 16 <= CAP): slow, but it terminates and is correct for adversarial orders and for
// massive ties (ties are ordered by index, so an exact prune always reaches KK).
// After the stream each row is cut to [KK, M], bitonic-sorted by its wave in registers and ranks 0, d, 2d, .. emitted.
#pragma once

constexpr int KF_WAVES = 4;
constexpr int KF_Q = 16;        // query rows per workgroup (one 16-column MFMA tile)
constexpr int KF_STEP = 32;     // candidates per wave and step: two 16-row MFMA tiles (two independent accumulators)

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct KnnFusedArgs {
    const float *xp;   // [rows / 16][4][64] float4, operand order (see above); rows = B * N
    const float *sq;   // [rows] squared norms in torch.sum's order
    int32_t *out;      // [rows][k]
    int N;             // points per room, multiple of 16
    int k, d;
    int KK;            // (k - 1) d + 1 keys decide the output
    int M;             // sort size: power of two >= KK, <= CAP
    int HIGH, LOW, TOL;   // prune when an append leaves a row above HIGH; rows above LOW are cut to [KK, KK + TOL]
};

// xp / sq for the stand-alone entry point: one wave per point, lane = feature (C = 64)
__global__ void knn_prep_kernel(const float *__restrict__ x, int ld, size_t rows, float *__restrict__ xp, float *__restrict__ sq)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t v = t >> 6;
    if (v >= rows) return;
    const int c = (int)(t & 63);
    const float val = x[v * ld + c];
    const int s = c >> 2, g = c & 3;
    xp[(v >> 4) * 1024 + (size_t)((((s >> 2) * 64) + (int)(v & 15) + 16 * g) * 4 + (s & 3))] = val;
    if (sq) {
        // torch.sum(x * x, -1) for 64 contiguous floats (same order as sumsq_rows_kernel / edge_max_fwd_kernel)
        const int lane = threadIdx.x & 63;
        const float q2 = __fmul_rn(val, val);
        const float u = __fadd_rn(q2, __shfl(q2, (lane + 32) & 63));
        float tl = __fadd_rn(u, __shfl(u, (lane + 8) & 63));
        tl = __fadd_rn(tl, __shfl(u, (lane + 16) & 63));
        tl = __fadd_rn(tl, __shfl(u, (lane + 24) & 63));
        float sacc = __shfl(tl, 0);
#pragma unroll
        for (int l = 1; l < 8; ++l) sacc = __fadd_rn(sacc, __shfl(tl, l));
        if (lane == 0) sq[v] = sacc;
    }
}

__device__ __forceinline__ unsigned long long wave_minmax_u64(unsigned long long v, bool want_max)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = ((unsigned long long)__shfl_xor((unsigned)(v >> 32), m) << 32) | __shfl_xor((unsigned)v, m);
        v = want_max ? (o > v ? o : v) : (o < v ? o : v);
    }
    return v;
}

// Cut row `ent` (T composite keys, T <= 64 * NPL) to the C smallest with KK <= C <= KK + tol (tol = 0: exactly KK),
// compacted in place; returns C and the threshold (every kept key <= thr, every dropped key > thr).  One wave.
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
    unsigned long long mn = v[0], mx = 0ull;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        mn = v[i] < mn ? v[i] : mn;
        mx = (v[i] != ~0ull && v[i] > mx) ? v[i] : mx;
    }
    mn = wave_minmax_u64(mn, false);
    mx = wave_minmax_u64(mx, true);
    // invariant: #(v <= lo) < KK <= #(v <= hi) = c_hi
    unsigned long long lo = mn - 1ull, hi = mx;
    unsigned c_hi = T;
    while (c_hi > KK + tol && hi - lo > 1ull) {
        // midpoint in DISTANCE space (the integer keys between a ~0 self distance and the populated binades would
        // cost ~10 extra halvings), all indices admitted at that distance; integer midpoint when that does not split
        const float dm = 0.5f * dist_of((unsigned)(lo >> 12)) + 0.5f * dist_of((unsigned)(hi >> 12));
        unsigned long long mid = ((unsigned long long)key_of(dm) << 12) | 0xFFFull;
        if (!(mid > lo && mid < hi)) mid = lo + ((hi - lo) >> 1);
        unsigned c = 0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) c += v[i] <= mid ? 1u : 0u;
        c = wave_sum_u32(c);
        if (c < KK) lo = mid;
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

template <int NPL>
__global__ __launch_bounds__(KF_WAVES * 64) void knn_fused_kernel(KnnFusedArgs a)
{
    constexpr int CAP = NPL == 10 ? 624 : 64 * NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char kf_smem[];
    unsigned long long *ent = (unsigned long long *)kf_smem;                 // [KF_Q][CAP]
    unsigned long long *tau = ent + KF_Q * CAP;                              // [KF_Q]
    unsigned *cnt = (unsigned *)(tau + KF_Q);                                // [KF_Q]
    unsigned *flag = cnt + KF_Q;                                             // [3][2]: overflow, prune (by step % 3)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, g = lane >> 4;
    const int tiles_per_room = a.N >> 4;
    const int room = blockIdx.x / tiles_per_room;
    const size_t room_row0 = (size_t)room * a.N;
    const size_t tile0 = (size_t)room * tiles_per_room;                      // first operand tile of the room
    const size_t qrow = (size_t)blockIdx.x * KF_Q + q;
    const unsigned KK = (unsigned)a.KK;

    if (tid < KF_Q) { tau[tid] = ~0ull - 1ull; cnt[tid] = 0u; }   // (~0ull marks a candidate beyond N: never admitted)
    if (tid < 6) flag[tid] = 0u;

    const float4 *xp4 = (const float4 *)a.xp;
    const float4 *sq4 = (const float4 *)(a.sq + room_row0);
    float4 bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = xp4[((size_t)blockIdx.x * 4 + i) * 64 + lane];
    const float sqi = a.sq[qrow];

    const int n_pairs = (a.N + KF_STEP - 1) / KF_STEP;                       // candidate pairs-of-tiles in the room
    const int n_steps = (n_pairs + KF_WAVES - 1) / KF_WAVES;
    const int last_tile = tiles_per_room - 1;

    float4 an0[4], an1[4], sn0, sn1;
    auto load_step = [&](int step) {
        const int pair = step * KF_WAVES + wave;
        const int t0 = min(2 * pair, last_tile), t1 = min(2 * pair + 1, last_tile);   // clamped: masked out below
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            an0[i] = xp4[((tile0 + t0) * 4 + i) * 64 + lane];
            an1[i] = xp4[((tile0 + t1) * 4 + i) * 64 + lane];
        }
        sn0 = sq4[t0 * 4 + g];
        sn1 = sq4[t1 * 4 + g];
    };
    load_step(0);
    __syncthreads();

    for (int step = 0; step < n_steps; ++step) {
        float4 a0[4], a1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a0[i] = an0[i]; a1[i] = an1[i]; }
        const float4 s0 = sn0, s1 = sn1;
        if (step + 1 < n_steps) load_step(step + 1);

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

        // ---- composite keys of the lane's 8 (query, candidate) pairs
        const int pair = step * KF_WAVES + wave;
        const int c0 = pair * KF_STEP + 4 * g;              // candidates c0 + r (tile 0) and c0 + 16 + r (tile 1)
        unsigned long long comp[8];
        const float sj[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float z = r < 4 ? acc0[r] : acc1[r - 4];
            const int c = c0 + (r < 4 ? r : 12 + r);
            // (|x_i|^2 + (-2 x_i.x_j)) + |x_j|^2, torch_edge.py:41-43 (-2 z is exact, so the fma rounds once like the add)
            const float dist = __fadd_rn(__fmaf_rn(-2.0f, z, sqi), sj[r]);
            comp[r] = c < a.N ? (((unsigned long long)key_of(dist) << 12) | (unsigned)c) : ~0ull;
        }
        const unsigned long long tq = tau[q];
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) mask |= (comp[r] <= tq ? 1u : 0u) << r;
        const unsigned n = (unsigned)__popc(mask);
        unsigned base = 0;
        if (n) base = atomicAdd(&cnt[q], n);
        const int ph = step % 3;
        if (__ballot(n && base + n > (unsigned)CAP) && lane == 0) flag[2 * ph] = 1u;
        if (__ballot(n && base + n > (unsigned)a.HIGH) && lane == 0) flag[2 * ph + 1] = 1u;
        __syncthreads();
        const unsigned f_ovf = flag[2 * ph], f_prune = flag[2 * ph + 1];
        if (tid == 0) { flag[2 * ((step + 2) % 3)] = 0u; flag[2 * ((step + 2) % 3) + 1] = 0u; }

        if (!f_ovf) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if ((mask >> r) & 1u) ent[q * CAP + base + __popc(mask & ((1u << r) - 1u))] = comp[r];
            if (f_prune) {
                __syncthreads();
                for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
                    const unsigned T = cnt[qq];
                    if (T > (unsigned)a.LOW) {
                        unsigned long long thr;
                        const unsigned c = prune_row<NPL>(ent + qq * CAP, T, KK, (unsigned)a.TOL, lane, thr);
                        if (lane == 0) { cnt[qq] = c; tau[qq] = thr; }
                    }
                }
                __syncthreads();
            }
        } else {
            // ---- overflow (rare): roll the step back, cut every row to exactly KK, commit one register at a time
            if (n) atomicSub(&cnt[q], n);
            __syncthreads();
            for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
                const unsigned T = cnt[qq];
                if (T > KK) {
                    unsigned long long thr;
                    const unsigned c = prune_row<NPL>(ent + qq * CAP, T, KK, 0u, lane, thr);
                    if (lane == 0) { cnt[qq] = c; tau[qq] = thr; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool pass = comp[r] <= tau[q];
                if (pass) ent[q * CAP + atomicAdd(&cnt[q], 1u)] = comp[r];     // <= 16 per row: KK + 16 <= CAP
                __syncthreads();
                for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
                    const unsigned T = cnt[qq];
                    if (T + 16u > (unsigned)CAP) {
                        unsigned long long thr;
                        const unsigned c = prune_row<NPL>(ent + qq * CAP, T, KK, 0u, lane, thr);
                        if (lane == 0) { cnt[qq] = c; tau[qq] = thr; }
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();

    // ---- final order: cut to [KK, M], pad, sort, emit every d-th
    const unsigned M = (unsigned)a.M;
    for (int qq = wave; qq < KF_Q; qq += KF_WAVES) {
        unsigned long long *row = ent + qq * CAP;
        unsigned T = cnt[qq];
        if (T > M) {
            unsigned long long thr;
            T = prune_row<NPL>(row, T, KK, M - KK, lane, thr);
        }
        wave_lds_fence();
        for (unsigned t = T + lane; t < M; t += 64) row[t] = ~0ull;
        wave_lds_fence();
        switch (M) {
        case 64: wave_sort_keys<1>(row, lane); break;
        case 128: wave_sort_keys<2>(row, lane); break;
        case 256: wave_sort_keys<4>(row, lane); break;
        default: wave_sort_keys<8>(row, lane); break;
        }
        wave_lds_fence();
        if (lane < a.k) a.out[((size_t)blockIdx.x * KF_Q + qq) * a.k + lane] = (int32_t)(row[(size_t)lane * a.d] & 0xFFFull);
    }
}

inline size_t knn_fused_lds_bytes(int npl)
{
    const int cap = npl == 10 ? 624 : 64 * npl;
    return (size_t)KF_Q * cap * 8 + KF_Q * 8 + KF_Q * 4 + 6 * 4;
}

// class of a dilation: the smallest buffer with room for M and a useful amount of slack
inline int knn_fused_class(int KK)
{
    if (KK <= 46) return 2;      // CAP 128
    if (KK <= 136) return 4;     // CAP 256
    if (KK <= 256) return 6;     // CAP 384
    return 10;                   // CAP 624 (KK <= 406 + ...)
}
