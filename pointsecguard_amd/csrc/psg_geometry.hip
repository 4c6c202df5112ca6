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
    // many problems (a whole attack plan: iterations x rooms): 256 threads x 16 points keeps three problems per CU
    // resident and needs a 4-wave instead of a 16-wave reduction per step (3.1 -> 2.4 ms for 1280 problems);
    // few problems: 1024 threads x 4 points minimises the latency of each step
    if (N <= 4096 && P > 512) return launch_fps<256, 16>(xyz, n_clouds, P, N, S, start, out_idx, st);
    if (N <= 4096) return launch_fps<1024, 4>(xyz, n_clouds, P, N, S, start, out_idx, st);
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
    size_t lds = (size_t)((N + 255) & ~255) * 4 * sizeof(float);
    if (lds > 48 * 1024)
        PSG_CHECK_HIP(hipFuncSetAttribute((const void *)ball_query_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
    hipLaunchKernelGGL(ball_query_kernel, dim3(psg::ceil_div(S, BQ_CPB), P), dim3(BQ_THREADS), lds,
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
