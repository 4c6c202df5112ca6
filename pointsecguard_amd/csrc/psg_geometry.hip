// Geometry kernels of the PointNet++ hot path for gfx950: farthest-point sampling, ball query,
// 3-NN + inverse-distance weights, row gather.  All of them are integer-output (or feed integer
// decisions), so the fp32 evaluation order is the one pinned in SURVEY.md section 8(a') and every
// multiply/add below is written with explicit rounding intrinsics (this file is also compiled with
// -ffp-contract=off).  VALU fp32 on purpose: MFMA accumulation order would break bit-exact grouping.
//
// Reference semantics (paths relative to /root/reference):
//   farthest_point_sample  PointNet/models/pointnet_util.py:63-84
//   query_ball_point       PointNet/models/pointnet_util.py:87-107  (square_distance :19-40)
//   3-NN interpolation     PointNet/models/pointnet_util.py:301-307
//   index_points           PointNet/models/pointnet_util.py:43-60
//
// Batching: every kernel runs P independent "problems" (= rooms x attack iterations).  A problem p
// reads cloud (p % n_clouds) of a [n_clouds][N][3] array, so the level-0 xyz of a room is shared by
// all attack iterations while deeper levels have one cloud per problem.
#include "psg_common.h"

namespace {

__device__ __forceinline__ float sumsq3(float x, float y, float z)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}

// square_distance(src, dst) for one pair: ((-2*dot) + |src|^2) + |dst|^2, dot = FMA chain over k
__device__ __forceinline__ float sqdist(float sx, float sy, float sz, float ssq, float dx, float dy, float dz,
                                        float dsq)
{
    float dot = __fmaf_rn(sz, dz, __fmaf_rn(sy, dy, __fmul_rn(sx, dx)));
    return __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), ssq), dsq);
}

typedef float v2f __attribute__((ext_vector_type(2)));   // operand of the packed-fp32 pipe (v_pk_*_f32)

// ---- wave-level max with DPP (no LDS round trips) --------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v)
{
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
    return o > v ? o : v;
}
// max over each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ unsigned row_max_u32(unsigned v)
{
    v = dpp_max_u32<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_max_u32<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_max_u32<0x141, 0xF>(v);  // row_half_mirror
    v = dpp_max_u32<0x140, 0xF>(v);  // row_mirror
    return v;
}
// max over the 64 lanes of the wave, returned wave-uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = row_max_u32(v);
    v = dpp_max_u32<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v = dpp_max_u32<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---------------------------------------------------------------------------------------------
// FPS: one workgroup per problem; thread t owns the PPT consecutive points PPT*t .. PPT*t+PPT-1
// (so a lower lane / lower wave always means a lower point index).  xyz stays in LDS for the centroid
// broadcast and in registers for the distance update.  Per step:
//   distance update + in-thread argmax  ->  wave max by DPP  ->  ballot picks the lowest lane holding
//   it  ->  one LDS slot per wave, ONE barrier (slots double-buffered)  ->  16-lane DPP max over the
//   wave partials, ballot picks the lowest wave.
// Distances are non-negative floats, compared as their bit patterns (exact same order); ties resolve to
// the lowest index like torch.max on CPU.
// ---------------------------------------------------------------------------------------------
template <int NT, int PPT>
__global__ __launch_bounds__(NT) void fps_kernel(const float *__restrict__ xyz, int n_clouds, int N, int S,
                                                 const int32_t *__restrict__ start, int32_t *__restrict__ out)
{
    constexpr int NW = NT / 64;
    extern __shared__ float smem[];
    float *s_xyz = smem;                                 // [NT*PPT*3] (padded cloud)
    uint2 *s_part = (uint2 *)(smem + NT * PPT * 3);      // [2][NW] (value bits, index)

    const int p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *src = xyz + (size_t)(p % n_clouds) * N * 3;
    for (int i = tid; i < NT * PPT * 3; i += NT) s_xyz[i] = i < N * 3 ? src[i] : 0.0f;
    __syncthreads();

    // points in registers as pairs: the distance update runs on the packed-fp32 pipe (two points per instruction,
    // every lane result bit-identical to the scalar sub / mul / add sequence of sumsq3)
    constexpr int PP = (PPT + 1) / 2;
    v2f px[PP], py[PP], pz[PP], dist[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        const int i = tid * PPT + 2 * q, i1 = (2 * q + 1 < PPT) ? i + 1 : i;
        px[q] = v2f{s_xyz[3 * i], s_xyz[3 * i1]};
        py[q] = v2f{s_xyz[3 * i + 1], s_xyz[3 * i1 + 1]};
        pz[q] = v2f{s_xyz[3 * i + 2], s_xyz[3 * i1 + 2]};
        dist[q] = v2f{1e10f, 1e10f};
    }
    int far = start[p];
    int32_t *o = out + (size_t)p * S;
    for (int s = 0; s < S; ++s) {
        if (tid == 0) o[s] = far;
        const float cx = s_xyz[3 * far], cy = s_xyz[3 * far + 1], cz = s_xyz[3 * far + 2];
        const v2f cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
        unsigned best = 0u;
        int bq = 0;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const v2f dx = px[q] - cx2, dy = py[q] - cy2, dz = pz[q] - cz2;
            const v2f d = ((dx * dx) + (dy * dy)) + (dz * dz);
            dist[q] = v2f{d[0] < dist[q][0] ? d[0] : dist[q][0], d[1] < dist[q][1] ? d[1] : dist[q][1]};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (2 * q + u >= PPT) break;
                // points past N never win: their key is 0 and the real maximum is >= 0 at a lower index
                unsigned key = (tid * PPT + 2 * q + u) < N ? __float_as_uint(dist[q][u]) : 0u;
                if (key > best) { best = key; bq = 2 * q + u; }
            }
        }
        const unsigned wmax = wave_max_u32(best);
        const unsigned long long hit = __ballot(best == wmax);
        const int src_lane = __builtin_ctzll(hit);
        int widx = __builtin_amdgcn_readlane(tid * PPT + bq, src_lane);
        if (NW > 1) {
            uint2 *slot = s_part + (s & 1) * NW;
            if (lane == 0) slot[wave] = make_uint2(wmax, (unsigned)widx);
            __syncthreads();
            const uint2 part = slot[lane & (NW - 1)];
            const unsigned gmax = row_max_u32(part.x);
            const unsigned long long hw = __ballot(part.x == gmax) & ((1ull << NW) - 1ull);
            widx = __builtin_amdgcn_readlane((int)part.y, __builtin_ctzll(hw));
        }
        far = widx;
    }
}

// out[p][s][:] = points[p % n_clouds][idx[p][s]][:]
__global__ void gather_rows_kernel(const float *__restrict__ pts, int n_clouds, int N, int C,
                                   const int32_t *__restrict__ idx, int S, float *__restrict__ out, int P)
{
    size_t total = (size_t)P * S * C;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(t % C);
        size_t r = t / C;
        int s = (int)(r % S);
        int p = (int)(r / S);
        out[t] = pts[((size_t)(p % n_clouds) * N + idx[(size_t)p * S + s]) * C + c];
    }
}

// ---------------------------------------------------------------------------------------------
// Ball query: the cloud (x,y,z,|p|^2) is staged once per workgroup in LDS; one wave per centroid
// scans it 64 points at a time in index order, compacting hits with ballot + prefix popcount and
// stopping at K.  No [S,N] matrix, no sort (the reference materialises both).
// ---------------------------------------------------------------------------------------------
constexpr int BQ_THREADS = 1024;
constexpr int BQ_CPB = 128;  // centroids per workgroup (8 per wave)
constexpr int BQ_UNROLL = 4; // 64-point chunks in flight per wave (hides the LDS latency of the scan)

__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(const float *__restrict__ xyz, int n_clouds,
                                                                const float *__restrict__ new_xyz, int N, int S,
                                                                float r2, int K, int32_t *__restrict__ out)
{
    // structure-of-arrays staging: a lane's points j and j + 64 of one plane come back from ONE ds_read2_b32 in
    // adjacent registers, i.e. directly as an operand of the packed-fp32 pipe (two distances per instruction)
    extern __shared__ float s_soa[];   // x[NP] y[NP] z[NP] |p|^2[NP], NP = N rounded up to 256
    const int NP = (N + 255) & ~255;
    float *s_x = s_soa, *s_y = s_soa + NP, *s_z = s_soa + 2 * NP, *s_q = s_soa + 3 * NP;
    const int p = blockIdx.y;
    const float *src = xyz + (size_t)(p % n_clouds) * N * 3;
    for (int i = threadIdx.x; i < NP; i += BQ_THREADS) {
        float x = 0.f, y = 0.f, z = 0.f, q = INFINITY;       // padding points are infinitely far away
        if (i < N) { x = src[3 * i]; y = src[3 * i + 1]; z = src[3 * i + 2]; q = sumsq3(x, y, z); }
        s_x[i] = x; s_y[i] = y; s_z[i] = z; s_q[i] = q;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int c_end = min(S, (int)(blockIdx.x + 1) * BQ_CPB);
    for (int c = blockIdx.x * BQ_CPB + wave; c < c_end; c += BQ_THREADS / 64) {
        const float *cp = new_xyz + ((size_t)p * S + c) * 3;
        const float cx = cp[0], cy = cp[1], cz = cp[2];
        const float csq = sumsq3(cx, cy, cz);
        const v2f cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz}, cs2 = {csq, csq}, m2 = {-2.0f, -2.0f};
        int32_t *o = out + ((size_t)p * S + c) * K;
        int cnt = 0, first = N;
        for (int base = 0; base < N && cnt < K; base += 64 * BQ_UNROLL) {
            bool in[BQ_UNROLL];
#pragma unroll
            for (int u = 0; u < BQ_UNROLL; u += 2) {
                const int j = base + u * 64 + lane;             // and j + 64
                const v2f qx = {s_x[j], s_x[j + 64]}, qy = {s_y[j], s_y[j + 64]}, qz = {s_z[j], s_z[j + 64]},
                          qs = {s_q[j], s_q[j + 64]};
                // square_distance order: ((-2 * fma chain) + |centroid|^2) + |point|^2, as sqdist()
                v2f dot = cx2 * qx;
                dot = __builtin_elementwise_fma(cy2, qy, dot);
                dot = __builtin_elementwise_fma(cz2, qz, dot);
                const v2f dd = ((m2 * dot) + cs2) + qs;
                in[u] = !(dd[0] > r2);                          // padding: +inf > r2
                in[u + 1] = !(dd[1] > r2);
            }
#pragma unroll
            for (int u = 0; u < BQ_UNROLL; ++u) {
                unsigned long long m = __ballot(in[u]);
                if (m) {
                    if (cnt == 0) first = base + u * 64 + __builtin_ctzll(m);
                    int pos = cnt + __popcll(m & lt_mask);
                    if (in[u] && pos < K) o[pos] = base + u * 64 + lane;
                    cnt += __popcll(m);
                }
            }
        }
        // pad with the first hit (pointnet_util.py:104-106); an empty ball emits N (reference would fault)
        for (int pos = min(cnt, K) + lane; pos < K; pos += 64) o[pos] = first;
    }
}

// ---------------------------------------------------------------------------------------------
// Ball query through a uniform grid (round 4), for clouds of 2048 to 4096 points (SA level 0).  The scan above tests every centroid
// against every point (1024 x 4096 pairs per problem at level 0: the kernel sat at ~45 % of the vector peak and was 3 %
// of an attack); a ball of radius r holds a few dozen points.  Per workgroup: the cloud is staged as before, its
// bounding box is reduced, points are binned into cells of edge >= 1.05 r (LDS counting sort: histogram, scan,
// scatter; the order inside a cell is arbitrary), and a wave then serves a centroid from the 27 cells around it
// (9 contiguous runs of the sorted index list): the candidates - typically 40 to 150 instead of 4096 - get the SAME
// distance arithmetic as the scan (sqdist: the reference's expansion, whose rounding decides membership at the ball
// boundary), hits set their bit in a per-wave bitmap over the N points, and the first K set bits ARE the reference's
// "first nsample indices in ascending order" (pointnet_util.py:100-106).  Pruning can only drop points that the exact
// test would reject: the expansion's rounding error is below 2.4e-7 (|c| + |p|)^2 <= 3e-6 max|coordinate|^2, so a point
// that passes the test has a true squared distance below r^2 + 1e-5 max|coordinate|^2 - and the cell edge is 1.05 x the
// root of exactly that, i.e. the point is less than one cell edge away per axis - and both cell indices come from the same
// monotone formula (a centroid outside the cloud's box is clamped to the border cell: what it can reach lies in the border
// cells).  A cloud with non-finite coordinates is one cell, i.e. the full scan.  The argument above is about centroids that
// can REACH the cloud: psg_ball_query is a public entry that takes arbitrary new_xyz, so a centroid that is non-finite
// (every `dd > r2` is false for a NaN: the scan returns 0 .. K-1) or lies more than one cell edge outside the cloud's box
// (far enough away, cancellation in the expansion passes points no cell walk would visit) takes ALL N points as its
// candidates - the scan, inside this kernel (advisor, round 4; tests/test_gpu_ball_grid.py).  Same indices, bit for bit
// (tests/test_gpu_parity.py: reference fixtures; test_unit_geometry_vs_oracle; tests/test_gpu_edge.py).
// ---------------------------------------------------------------------------------------------
constexpr int BQG_CELLS = 4096;     // cells per cloud at most (the edge grows by 1.25x until the box fits)
constexpr int BQG_CPB = 1024;       // centroids per workgroup (64 per wave): the grid build is shared by all queries of a level-0 problem

// inclusive prefix sum over the 64 lanes on the DPP network (row_shr 1 / 2 / 4 / 8, then row_bcast15 / row_bcast31); lanes
// shifted in from outside a row read 0.  (A __shfl_up scan is six dependent LDS round trips: three of them per centroid
// were most of this kernel's first version.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned bq_dpp0(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned bq_wave_incl_scan(unsigned v)
{
    v += bq_dpp0<0x111, 0xF>(v);
    v += bq_dpp0<0x112, 0xF>(v);
    v += bq_dpp0<0x114, 0xF>(v);
    v += bq_dpp0<0x118, 0xF>(v);
    v += bq_dpp0<0x142, 0xA>(v);
    v += bq_dpp0<0x143, 0xC>(v);
    return v;
}

__global__ __launch_bounds__(BQ_THREADS) void ball_query_grid_kernel(const float *__restrict__ xyz, int n_clouds,
                                                                     const float *__restrict__ new_xyz, int N, int S,
                                                                     float r2, int K, int32_t *__restrict__ out)
{
    extern __shared__ float s_soa[];   // x[NP] y[NP] z[NP] |p|^2[NP] | cell starts [CELLS + 1] | cursors [CELLS] = bitmaps | sorted [NP] u16
    const int NP = (N + 255) & ~255;
    float *s_x = s_soa, *s_y = s_soa + NP, *s_z = s_soa + 2 * NP, *s_q = s_soa + 3 * NP;
    int *s_start = (int *)(s_soa + 4 * NP);
    int *s_cur = s_start + BQG_CELLS + 1;
    unsigned short *s_sorted = (unsigned short *)(s_cur + BQG_CELLS);
    __shared__ float s_red[BQ_THREADS / 64][6];
    __shared__ float s_box[8];          // min x, y, z, 1 / cell edge; max x, y, z, cell edge
    __shared__ int s_dim[4];            // nx, ny, nz, cells
    __shared__ int s_wsum[BQ_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = blockIdx.y;
    const float *src = xyz + (size_t)(p % n_clouds) * N * 3;
    float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY;
    for (int i = tid; i < NP; i += BQ_THREADS) {
        float x = 0.f, y = 0.f, z = 0.f, q = INFINITY;
        if (i < N) {
            x = src[3 * i]; y = src[3 * i + 1]; z = src[3 * i + 2]; q = sumsq3(x, y, z);
            mn0 = fminf(mn0, x); mn1 = fminf(mn1, y); mn2 = fminf(mn2, z);
            mx0 = fmaxf(mx0, x); mx1 = fmaxf(mx1, y); mx2 = fmaxf(mx2, z);
        }
        s_x[i] = x; s_y[i] = y; s_z[i] = z; s_q[i] = q;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn0 = fminf(mn0, __shfl_xor(mn0, o)); mn1 = fminf(mn1, __shfl_xor(mn1, o)); mn2 = fminf(mn2, __shfl_xor(mn2, o));
        mx0 = fmaxf(mx0, __shfl_xor(mx0, o)); mx1 = fmaxf(mx1, __shfl_xor(mx1, o)); mx2 = fmaxf(mx2, __shfl_xor(mx2, o));
    }
    if (lane == 0) { s_red[wave][0] = mn0; s_red[wave][1] = mn1; s_red[wave][2] = mn2; s_red[wave][3] = mx0; s_red[wave][4] = mx1; s_red[wave][5] = mx2; }
    __syncthreads();
    if (tid == 0) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int w = 0; w < BQ_THREADS / 64; ++w)
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], s_red[w][a]); hi[a] = fmaxf(hi[a], s_red[w][3 + a]); }
        float maxabs = 0.0f;
        for (int a = 0; a < 3; ++a) maxabs = fmaxf(maxabs, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
        maxabs += sqrtf(r2);                                           // (a centroid that reaches the cloud is at most r outside its box)
        float edge = 1.05f * sqrtf(r2 + 1e-5f * maxabs * maxabs);
        const bool finite = maxabs < 1e18f;                            // (false for NaN and infinities)
        int nx, ny, nz;
        for (;;) {
            if (!finite) { nx = ny = nz = 1; edge = 1.0f; break; }
            nx = (int)((hi[0] - lo[0]) / edge) + 1; ny = (int)((hi[1] - lo[1]) / edge) + 1; nz = (int)((hi[2] - lo[2]) / edge) + 1;
            if ((long long)nx * ny * nz <= BQG_CELLS) break;
            edge *= 1.25f;
        }
        s_box[0] = lo[0]; s_box[1] = lo[1]; s_box[2] = lo[2]; s_box[3] = 1.0f / edge;
        s_box[4] = hi[0]; s_box[5] = hi[1]; s_box[6] = hi[2]; s_box[7] = edge;
        s_dim[0] = nx; s_dim[1] = ny; s_dim[2] = nz; s_dim[3] = nx * ny * nz;
    }
    __syncthreads();
    const float bx = s_box[0], by = s_box[1], bz = s_box[2], inv = s_box[3];
    const float hx = s_box[4], hy = s_box[5], hz = s_box[6], edge1 = s_box[7];
    const int nx = s_dim[0], ny = s_dim[1], nz = s_dim[2], ncell = s_dim[3];
    auto cell1 = [&](float v, float b, int n) {
        if (n == 1) return 0;
        const int c = (int)((v - b) * inv);
        return c < 0 ? 0 : (c >= n ? n - 1 : c);
    };
    for (int c = tid; c <= ncell; c += BQ_THREADS) s_start[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += BQ_THREADS)
        atomicAdd(&s_start[(cell1(s_z[i], bz, nz) * ny + cell1(s_y[i], by, ny)) * nx + cell1(s_x[i], bx, nx) + 1], 1);
    __syncthreads();
    {   // inclusive scan of s_start[1 .. ncell] in place (s_start[c + 1] = end of cell c): four cells per thread
        const int c0 = 1 + 4 * tid;
        int v[4], sum = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = c0 + u <= ncell ? s_start[c0 + u] : 0; sum += v[u]; v[u] = sum; }
        const int inc = (int)bq_wave_incl_scan((unsigned)sum);
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        int base = inc - sum;
        for (int w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + u <= ncell) s_start[c0 + u] = base + v[u];
    }
    __syncthreads();
    for (int c = tid; c < ncell; c += BQ_THREADS) s_cur[c] = s_start[c];
    __syncthreads();
    for (int i = tid; i < N; i += BQ_THREADS) {
        const int cell = (cell1(s_z[i], bz, nz) * ny + cell1(s_y[i], by, ny)) * nx + cell1(s_x[i], bx, nx);
        s_sorted[atomicAdd(&s_cur[cell], 1)] = (unsigned short)i;
    }
    __syncthreads();
    // ---- queries: one wave per centroid; the cursors are dead, their memory holds the per-wave bitmaps
    const int words = NP >> 5;
    unsigned *bm = (unsigned *)s_cur + wave * words;
    const int c_end = min(S, (int)(blockIdx.x + 1) * BQG_CPB);
    for (int c = blockIdx.x * BQG_CPB + wave; c < c_end; c += BQ_THREADS / 64) {
        for (int w = lane; w < words; w += 64) bm[w] = 0u;
        const float *cp = new_xyz + ((size_t)p * S + c) * 3;
        const float cx = cp[0], cy = cp[1], cz = cp[2];
        const float csq = sumsq3(cx, cy, cz);
        const int ix = cell1(cx, bx, nx), iy = cell1(cy, by, ny), iz = cell1(cz, bz, nz);
        // the 9 runs of the sorted list: rows (iz + dz, iy + dy), cells ix - 1 .. ix + 1 (contiguous: x runs fastest)
        int start = 0, len = 0;
        // (written so that a NaN coordinate makes it true)
        const bool off_cloud = !(cx >= bx - edge1 && cx <= hx + edge1 && cy >= by - edge1 && cy <= hy + edge1 && cz >= bz - edge1 &&
                                 cz <= hz + edge1);
        if (off_cloud) {
            len = lane == 0 ? N : 0;      // one run: the whole sorted list = every point of the cloud
        } else if (lane < 9) {
            const int z = iz + lane / 3 - 1, y = iy + lane % 3 - 1;
            if (z >= 0 && z < nz && y >= 0 && y < ny) {
                const int x0 = ix > 0 ? ix - 1 : 0, x1 = ix + 1 < nx ? ix + 1 : nx - 1;
                const int lin = (z * ny + y) * nx + x0;
                start = s_start[lin];
                len = s_start[lin + (x1 - x0) + 1] - start;
            }
        }
        const int incl = (int)bq_wave_incl_scan((unsigned)len);
        const int T = __builtin_amdgcn_readlane(incl, 8);
        const int off = incl - len;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i0 = 0; i0 < T; i0 += 64) {
            const int i = i0 + lane;
            int pos = -1;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int o = __builtin_amdgcn_readlane(off, r), l = __builtin_amdgcn_readlane(len, r),
                          st = __builtin_amdgcn_readlane(start, r);
                pos = (i >= o && i < o + l) ? st + (i - o) : pos;
            }
            if (pos >= 0) {
                const int j = s_sorted[pos];
                const float dd = sqdist(cx, cy, cz, csq, s_x[j], s_y[j], s_z[j], s_q[j]);
                if (!(dd > r2)) atomicOr(&bm[j >> 5], 1u << (j & 31));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the first K set bits, ascending; pad with the first hit (pointnet_util.py:104-106); an empty ball emits N
        int32_t *o = out + ((size_t)p * S + c) * K;
        int cnt = 0, first = N;
        for (int w0 = 0; w0 < words && cnt < K; w0 += 64) {
            unsigned word = w0 + lane < words ? bm[w0 + lane] : 0u;
            const int pc = __popc(word);
            const int inc = (int)bq_wave_incl_scan((unsigned)pc);
            const int tot = __builtin_amdgcn_readlane(inc, 63);
            if (tot == 0) continue;
            if (cnt == 0) {
                const int fl = __builtin_ctzll(__ballot(word != 0u));
                first = (w0 + fl) * 32 + __builtin_ctz((unsigned)__builtin_amdgcn_readlane((int)word, fl));
            }
            int rank = cnt + inc - pc;
            while (word && rank < K) {
                o[rank++] = (w0 + lane) * 32 + __builtin_ctz(word);
                word &= word - 1u;
            }
            cnt += tot;
        }
        for (int pos = min(cnt, K) + lane; pos < K; pos += 64) o[pos] = first;
        __builtin_amdgcn_wave_barrier();
    }
}
// ---------------------------------------------------------------------------------------------
// 3-NN: one thread per fine point, coarse cloud broadcast from LDS, top-3 kept in registers with
// strict '<' so equal distances keep the lower index first (stable ascending order).
// ---------------------------------------------------------------------------------------------
constexpr int NN_THREADS = 256;

// Two coarse points per step on the packed-fp32 pipe (v_pk_mul/fma/add_f32: two IEEE fp32 results per lane per
// instruction, each bit-identical to the scalar op): the coarse cloud is staged in LDS as pairs
// {x0,x1,y0,y1} {z0,z1,|p0|^2,|p1|^2}; an odd tail is padded with a point at infinite distance.
__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(const float *__restrict__ xyz1, int n_clouds1,
                                                              const float *__restrict__ xyz2, int N, int S,
                                                              int32_t *__restrict__ idx, float *__restrict__ w)
{
    extern __shared__ float4 s_pair[];  // [2 * ceil(S/2)]
    const int p = blockIdx.y;
    const float *c2 = xyz2 + (size_t)p * S * 3;
    const int S2 = (S + 1) >> 1;
    for (int i = threadIdx.x; i < S2; i += NN_THREADS) {
        const int a = 2 * i, b = 2 * i + 1;
        const float xa = c2[3 * a], ya = c2[3 * a + 1], za = c2[3 * a + 2];
        float xb = 0.f, yb = 0.f, zb = 0.f, sb = INFINITY;
        if (b < S) { xb = c2[3 * b]; yb = c2[3 * b + 1]; zb = c2[3 * b + 2]; sb = sumsq3(xb, yb, zb); }
        s_pair[2 * i] = make_float4(xa, xb, ya, yb);
        s_pair[2 * i + 1] = make_float4(za, zb, sumsq3(xa, ya, za), sb);
    }
    __syncthreads();
    const int i = blockIdx.x * NN_THREADS + threadIdx.x;
    if (i >= N) return;
    const float *fp = xyz1 + ((size_t)(p % n_clouds1) * N + i) * 3;
    const float fx = fp[0], fy = fp[1], fz = fp[2];
    const float fsq = sumsq3(fx, fy, fz);
    const v2f fx2 = {fx, fx}, fy2 = {fy, fy}, fz2 = {fz, fz}, fs2 = {fsq, fsq}, m2 = {-2.0f, -2.0f};
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    for (int jj = 0; jj < S2; ++jj) {
        const float4 q0 = s_pair[2 * jj], q1 = s_pair[2 * jj + 1];
        const v2f qx = {q0.x, q0.y}, qy = {q0.z, q0.w}, qz = {q1.x, q1.y}, qs = {q1.z, q1.w};
        // square_distance order: ((-2 * (fma chain over x, y, z)) + |fine|^2) + |coarse|^2
        v2f dot = fx2 * qx;
        dot = __builtin_elementwise_fma(fy2, qy, dot);
        dot = __builtin_elementwise_fma(fz2, qz, dot);
        const v2f dd = ((m2 * dot) + fs2) + qs;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float d = dd[u];
            const int j = 2 * jj + u;
            if (d < d2) {
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                    else { d1 = d; i1 = j; }
                } else { d2 = d; i2 = j; }
            }
        }
    }
    float r0 = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f));
    float r1 = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f));
    float r2 = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
    float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
    size_t o = ((size_t)p * N + i) * 3;
    idx[o] = i0; idx[o + 1] = i1; idx[o + 2] = i2;
    w[o] = __fdiv_rn(r0, norm); w[o + 1] = __fdiv_rn(r1, norm); w[o + 2] = __fdiv_rn(r2, norm);
}

// square_distance(src, dst) materialised (public helper of pointnet_util; not on the attack path)
__global__ void square_distance_kernel(const float *__restrict__ src, const float *__restrict__ dst, int N, int M,
                                       float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int i = blockIdx.y;
    const float *s = src + ((size_t)b * N + i) * 3;
    const float sx = s[0], sy = s[1], sz = s[2];
    const float ssq = sumsq3(sx, sy, sz);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
        const float *d = dst + ((size_t)b * M + j) * 3;
        out[((size_t)b * N + i) * M + j] = sqdist(sx, sy, sz, ssq, d[0], d[1], d[2], sumsq3(d[0], d[1], d[2]));
    }
}

template <int NT, int PPT>
int launch_fps(const float *xyz, int n_clouds, int P, int N, int S, const int32_t *start, int32_t *out,
               hipStream_t st)
{
    size_t lds = (size_t)NT * PPT * 3 * 4 + 2 * (NT / 64) * 8;
    if (lds > 48 * 1024)
        PSG_CHECK_HIP(hipFuncSetAttribute((const void *)fps_kernel<NT, PPT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((fps_kernel<NT, PPT>), dim3(P), dim3(NT), lds, st, xyz, n_clouds, N, S, start, out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

}  // namespace

extern "C" int psg_fps(psg_ctx *ctx, const float *xyz, int n_clouds, int P, int N, int S, const int32_t *start,
                       int32_t *out_idx, psg_stream stream)
{
    PSG_REQUIRE(ctx && xyz && start && out_idx, "psg_fps: null argument");
    PSG_REQUIRE(P > 0 && n_clouds > 0 && N > 0 && S > 0 && S <= N, "psg_fps: bad sizes P=%d N=%d S=%d", P, N, S);
    PSG_REQUIRE(N <= 8192, "psg_fps: N=%d exceeds the LDS-resident limit 8192", N);
    hipStream_t st = (hipStream_t)stream;
    if (N <= 64) return launch_fps<64, 1>(xyz, n_clouds, P, N, S, start, out_idx, st);
    if (N <= 256) return launch_fps<64, 4>(xyz, n_clouds, P, N, S, start, out_idx, st);
    if (N <= 1024) return launch_fps<256, 4>(xyz, n_clouds, P, N, S, start, out_idx, st);
    // many problems (a whole attack plan: iterations x rooms): 256 threads x 16 points keeps three problems per CU resident and
    // needs a 4-wave instead of a 16-wave reduction per step; few problems: the latency of a step is what counts, and 512 threads
    // x 8 points has it (tools/fps_cfg_probe.py on MI355X, microseconds per 4096 -> 1024 call, 1024 x 4 / 512 x 8 / 256 x 16:
    // 1 - 128 problems 690 / 672 / 735, 320 problems 1167 / 1010 / 978, 640: 1733 / 1420 / 1324, 2560: 5254 / 4071 / 3798)
    static const int cfg = psg::env_int("PSG_FPS_CFG", 0);   // measurement switch: 1 = 1024 x 4, 2 = 512 x 8, 3 = 256 x 16
    if (N <= 4096 && cfg == 1) return launch_fps<1024, 4>(xyz, n_clouds, P, N, S, start, out_idx, st);
    if (N <= 4096 && (cfg == 3 || (cfg == 0 && P >= 256))) return launch_fps<256, 16>(xyz, n_clouds, P, N, S, start, out_idx, st);
    if (N <= 4096) return launch_fps<512, 8>(xyz, n_clouds, P, N, S, start, out_idx, st);
    return launch_fps<1024, 8>(xyz, n_clouds, P, N, S, start, out_idx, st);
}

extern "C" int psg_square_distance(psg_ctx *ctx, const float *src, const float *dst, int B, int N, int M, float *out,
                                   psg_stream stream)
{
    PSG_REQUIRE(ctx && src && dst && out, "psg_square_distance: null argument");
    PSG_REQUIRE(B > 0 && N > 0 && M > 0 && N <= 65535 && B <= 65535, "psg_square_distance: bad sizes");
    hipLaunchKernelGGL(square_distance_kernel, dim3(std::min(64, psg::ceil_div(M, 256)), N, B), dim3(256), 0,
                       (hipStream_t)stream, src, dst, N, M, out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_gather_points(psg_ctx *ctx, const float *points, int n_clouds, int P, int N, int C,
                                 const int32_t *idx, int S, float *out, psg_stream stream)
{
    PSG_REQUIRE(ctx && points && idx && out, "psg_gather_points: null argument");
    PSG_REQUIRE(P > 0 && n_clouds > 0 && N > 0 && C > 0 && S > 0, "psg_gather_points: bad sizes");
    size_t total = (size_t)P * S * C;
    int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, points, n_clouds, N, C,
                       idx, S, out, P);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_ball_query(psg_ctx *ctx, const float *xyz, int n_clouds, const float *new_xyz, int P, int N,
                              int S, float r2, int K, int32_t *out_idx, psg_stream stream)
{
    PSG_REQUIRE(ctx && xyz && new_xyz && out_idx, "psg_ball_query: null argument");
    PSG_REQUIRE(P > 0 && n_clouds > 0 && N > 0 && S > 0 && K > 0, "psg_ball_query: bad sizes");
    PSG_REQUIRE(N <= 8192, "psg_ball_query: N=%d exceeds the LDS-resident limit 8192", N);
    // Clouds of 2048 .. 4096 points: the grid kernel (same indices; see its header).  Measured per launch of 2560 problems
    // (tools/ball_query_probe.py): N = 4096, S = 1024: 1.44 ms against 2.87 ms for the scan; N = 1024, S = 256: 0.29 against
    // 0.27 (the build is not repaid), so smaller clouds keep the scan.  PSG_BALL_QUERY=scan forces the scan (A/B runs).
    static const bool scan_only = []() { const char *v = psg::env_str("PSG_BALL_QUERY"); return v && std::string(v) == "scan"; }();
    const size_t soa = (size_t)((N + 255) & ~255) * 4 * sizeof(float);
    if (!scan_only && N >= 2048 && N <= 4096 && r2 > 0.0f && r2 < 1e30f) {
        const size_t lds = soa + (size_t)(2 * BQG_CELLS + 1) * 4 + (size_t)((N + 255) & ~255) * 2;
        static bool big = false;
        if (!big) {
            PSG_CHECK_HIP(hipFuncSetAttribute((const void *)ball_query_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            big = true;
        }
        hipLaunchKernelGGL(ball_query_grid_kernel, dim3(psg::ceil_div(S, BQG_CPB), P), dim3(BQ_THREADS), lds, (hipStream_t)stream, xyz,
                           n_clouds, new_xyz, N, S, r2, K, out_idx);
        PSG_LAUNCH_CHECK();
        return PSG_OK;
    }
    if (soa > 48 * 1024)
        PSG_CHECK_HIP(hipFuncSetAttribute((const void *)ball_query_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)soa));
    hipLaunchKernelGGL(ball_query_kernel, dim3(psg::ceil_div(S, BQ_CPB), P), dim3(BQ_THREADS), soa,
                       (hipStream_t)stream, xyz, n_clouds, new_xyz, N, S, r2, K, out_idx);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_three_nn(psg_ctx *ctx, const float *xyz1, int n_clouds1, const float *xyz2, int P, int N, int S,
                            int32_t *out_idx, float *out_w, psg_stream stream)
{
    PSG_REQUIRE(ctx && xyz1 && xyz2 && out_idx && out_w, "psg_three_nn: null argument");
    PSG_REQUIRE(P > 0 && n_clouds1 > 0 && N > 0 && S >= 3, "psg_three_nn: bad sizes (need S >= 3)");
    PSG_REQUIRE(S <= 8192, "psg_three_nn: S=%d exceeds the LDS-resident limit 8192", S);
    size_t lds = (size_t)2 * ((S + 1) / 2) * sizeof(float4);
    if (lds > 48 * 1024)
        PSG_CHECK_HIP(hipFuncSetAttribute((const void *)three_nn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
    hipLaunchKernelGGL(three_nn_kernel, dim3(psg::ceil_div(N, NN_THREADS), P), dim3(NN_THREADS), lds,
                       (hipStream_t)stream, xyz1, n_clouds1, xyz2, N, S, out_idx, out_w);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}
