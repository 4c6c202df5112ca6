// Device k-nearest-neighbour search on 3-D points for the RandLA-Net input pipeline (SURVEY.md section 8f rank 3:
// "device kNN replacing the nanoflann/OpenMP pipeline").
//
// Replaces DataProcessing.knn_search (RandLA-Net/helper_tool.py:158-167 of the reference) =
// nearest_neighbors.knn_batch(support, query, k, omp=True) (utils/nearest_neighbors/knn.pyx:71-109 ->
// knn_.cxx:103-134: one nanoflann kd-tree per batch element, an exact k-NN query per point, results sorted by
// ascending squared distance).  The reference calls it 2 x num_layers times per batch from its tf.data map function
// (main_S3DIS.py:198-207) on 40 960-point clouds: neigh_idx = knn(xyz, xyz, 16), up_i = knn(sub_points, xyz, 1).
//
// MI355X design: exact brute force instead of a tree.  40 960^2 = 1.7e9 candidate pairs is ~15 G lane-operations,
// well under a millisecond of VALU time on 256 CUs, has no divergence, no pointer chasing and no build phase, and --
// unlike a tree traversal -- visits candidates in index order, which makes the result canonical: ascending
// (distance, index).  One lane owns one query and keeps its K best in registers (sorted, fully unrolled
// compare-exchange insertion; after the first few hundred candidates an insertion is rare: ~K ln(N/K) per query);
// support points stream through LDS in tiles that every lane reads at the same address (broadcast, conflict free);
// four waves share 64 queries and split the support points, so a single 40 960-point cloud is 2560 waves.
// Distances use nanoflann's arithmetic, ((dx*dx) + dy*dy) + dz*dz without contraction (nanoflann.hpp L2_Adaptor, the
// dim < 4 tail loop), so equal inputs give equal distances and therefore equal neighbour sets.
#include "psg_common.h"

using namespace psg;

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int KNN_Q = 64;        // queries per workgroup (one per lane; all waves of the workgroup share them)
constexpr int KNN_SEG = 4;       // waves per workgroup: wave w scans the w-th quarter of the support points
constexpr int KNN_TILE = 256;    // support points a wave stages per pass (SoA x | y | z: 3 KiB per wave)

// sorted insertion of (d, i) into the K best (ascending (distance, index): callers present candidates in ascending
// index order and insert only on strict improvement, so of equal distances the lower index stays in front)
template <int K>
__device__ __forceinline__ void knn_insert(float (&bd)[K], int (&bi)[K], float d, int i)
{
    if (d < bd[K - 1]) {
        bd[K - 1] = d; bi[K - 1] = i;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            if (bd[j] < bd[j - 1]) {
                const float td = bd[j]; bd[j] = bd[j - 1]; bd[j - 1] = td;
                const int ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
            }
        }
    }
}

// Workgroup = 64 queries x 4 waves.  The candidate loop never takes a long divergent path: 8 candidates per step as
// four packed-fp32 pairs (v_pk_add / v_pk_mul, no contraction: ((dx*dx) + dy*dy) + dz*dz exactly); a candidate below
// the lane's threshold (its K-th best as of the last flush) is only APPENDED (branch-free) to the lane's pending queue in LDS.  With
// 64 queries per wave some lane improves in almost every step (a lane sees ~K ln(N/K) improvements, but the wave sees
// their union), so inserting on the spot would put the ~75-instruction sorted insertion into every step; queued, the
// insertions run in a few dozen wave-wide flushes (when some queue could overflow) over the whole scan.  Queue order is
// index order and an insertion requires a strict improvement, so equal distances keep the lower index in front.
// The four partial lists are merged through LDS by (distance, index).
constexpr int KNN_QCAP = 16;     // pending candidates per lane between flushes (a step adds at most 8)

template <int K>
__global__ __launch_bounds__(KNN_Q * KNN_SEG) void knn_points_kernel(const float *__restrict__ support, const float *__restrict__ query,
                                                                     int N1, int N2, int k_out, int32_t *__restrict__ out)
{
    static_assert(K <= KNN_QCAP, "the merge reuses the queue slabs");
    __shared__ float s_pts[KNN_SEG][3][KNN_TILE];
    __shared__ float2 s_q[KNN_SEG][KNN_QCAP][KNN_Q];   // pending {distance, index bits} per lane; reused by the final merge
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * KNN_Q + lane;
    const float *sup = support + (size_t)b * N1 * 3;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (q < N2) {
        const float *p = query + ((size_t)b * N2 + q) * 3;
        qx = p[0]; qy = p[1]; qz = p[2];
    }
    const v2f qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = INFINITY; bi[j] = 0; }
    float thr = INFINITY;
    int cnt = 0;
    auto flush = [&]() {   // wave-uniform trip count: the fullest queue
        for (int j = 0; __any(j < cnt); ++j)
            if (j < cnt) { const float2 e = s_q[w][j][lane]; knn_insert<K>(bd, bi, e.x, __float_as_int(e.y)); }
        cnt = 0;
        thr = bd[K - 1];
    };
    const int seg = ((N1 + KNN_SEG - 1) / KNN_SEG + 7) & ~7;        // support points per wave
    const int lo = w * seg, hi = min(N1, lo + seg);
    float *sx = s_pts[w][0], *sy = s_pts[w][1], *sz = s_pts[w][2];
    for (int t0 = 0; t0 < seg; t0 += KNN_TILE) {                    // the same trip count in every wave (barriers)
        const int base = lo + t0;
        __syncthreads();
        for (int i = lane; i < KNN_TILE; i += 64) {
            const int g = base + i;
            float x = INFINITY, y = INFINITY, z = INFINITY;         // padding: distance +inf, never below a threshold
            if (g < hi) { const float *p = sup + (size_t)g * 3; x = p[0]; y = p[1]; z = p[2]; }
            sx[i] = x; sy[i] = y; sz[i] = z;
        }
        __syncthreads();
        const int nt = min(KNN_TILE, max(0, hi - base));
        for (int i = 0; i < nt; i += 8) {
            float d[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const v2f px = *(const v2f *)(sx + i + 2 * u), py = *(const v2f *)(sy + i + 2 * u), pz = *(const v2f *)(sz + i + 2 * u);
                const v2f dx = qx2 - px, dy = qy2 - py, dz = qz2 - pz;
                const v2f dd = (dx * dx + dy * dy) + dz * dz;
                d[2 * u] = dd.x; d[2 * u + 1] = dd.y;
            }
            // branch-free append: every candidate is written to the lane's next free slot, the slot is kept (cnt
            // advances) only if the candidate is below the threshold.  cnt <= QCAP - 8 here, so slot cnt stays in range.
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s_q[w][cnt][lane] = make_float2(d[u], __int_as_float(base + i + u));
                cnt += d[u] < thr ? 1 : 0;
            }
            if (__any(cnt > KNN_QCAP - 8)) flush();
        }
    }
    flush();
    // merge the four sorted partial lists of every query by (distance, index)
#pragma unroll
    for (int j = 0; j < K; ++j) s_q[w][j][lane] = make_float2(bd[j], __int_as_float(bi[j]));
    __syncthreads();
    if (w == 0 && q < N2) {
        int head[KNN_SEG] = {0, 0, 0, 0};
        int32_t *o = out + ((size_t)b * N2 + q) * k_out;
        for (int j = 0; j < k_out; ++j) {
            float best = INFINITY;
            int bidx = 0x7FFFFFFF, bs = 0;
#pragma unroll
            for (int s2 = 0; s2 < KNN_SEG; ++s2) {
                if (head[s2] < K) {
                    const float2 e = s_q[s2][head[s2]][lane];
                    const float dv = e.x;
                    const int iv = __float_as_int(e.y);
                    if (dv < best || (dv == best && iv < bidx)) { best = dv; bidx = iv; bs = s2; }
                }
            }
            o[j] = bidx;
#pragma unroll
            for (int s2 = 0; s2 < KNN_SEG; ++s2) head[s2] += (s2 == bs) ? 1 : 0;
        }
    }
}

}  // namespace

extern "C" int psg_knn_points(psg_ctx *ctx, const float *support, const float *query, int batch, int n_support,
                              int n_query, int k, int32_t *out_idx, psg_stream stream)
{
    PSG_REQUIRE(ctx && support && query && out_idx, "psg_knn_points: null argument");
    PSG_REQUIRE(batch > 0 && batch <= 65535 && n_support > 0 && n_query > 0, "psg_knn_points: bad sizes");
    PSG_REQUIRE(k >= 1 && k <= 16 && k <= n_support, "psg_knn_points: k=%d must be in [1, min(16, n_support=%d)]", k, n_support);
    const dim3 grid(ceil_div(n_query, KNN_Q), batch);
    hipStream_t st = (hipStream_t)stream;
    if (k == 1)
        hipLaunchKernelGGL(knn_points_kernel<1>, grid, dim3(KNN_Q * KNN_SEG), 0, st, support, query, n_support, n_query, k, out_idx);
    else if (k <= 8)
        hipLaunchKernelGGL(knn_points_kernel<8>, grid, dim3(KNN_Q * KNN_SEG), 0, st, support, query, n_support, n_query, k, out_idx);
    else
        hipLaunchKernelGGL(knn_points_kernel<16>, grid, dim3(KNN_Q * KNN_SEG), 0, st, support, query, n_support, n_query, k, out_idx);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}
