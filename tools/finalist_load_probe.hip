// Probe (GPU box, round 6): what does the exact-distance pass of the prefilter kNN kernel's final ranking cost, and does it
// get cheaper when the 256-byte feature row of a finalist is fetched by FOUR lanes in one instruction (64 contiguous
// bytes per instruction and finalist) instead of by one lane in sixteen?  psg_knn_bf.cuh: kb_final_rows gives every lane
// its own finalist: 64 lanes x 16 loads of 16 bytes, 64 different rows per wave instruction ("Round 5 on
// the kNN kernels": 30k cycles per workgroup at d = 20 for ~2 800 finalists, three passes).  The ascending-k fmaf chain must
// stay one chain per finalist (bit-exactness against the reference's sgemm order), so cooperative loads need a 4 x 4
// transpose of the register blocks between the four lanes afterwards.
//   mode 0  one lane per finalist row (the kernel today)
//   mode 1  quads of ADJACENT lanes: lane g of a quad loads bytes [64 s + 16 g, + 16) of finalist phi's row for
//           (phi, s) in 4 x 4, then two butterfly stages of quad-permute selects (v_cndmask + DPP)
//   mode 2  groups of lanes {c, c + 16, c + 32, c + 48}: same loads with the row of 16 lanes as the piece index, transposed
//           with gfx950's v_permlane32_swap / v_permlane16_swap (one instruction per register pair)
//   mode 3 / 4  the loads of mode 1 / 2 WITHOUT the transpose (wrong results: timing bound of the memory side)
//   mode 5  mode 0's loads without the chain (memory side of today's pass)
// Every mode with a transpose must give the bits of mode 0 (checked on the host).
//   hipcc -O3 --offload-arch=gfx950 tools/finalist_load_probe.hip -o build/finalist_load_probe && build/finalist_load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned u2 __attribute__((ext_vector_type(2)));
constexpr int ROOM = 4096, ROOMS = 4, C = 64;

__device__ __forceinline__ unsigned hash_u32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <int CTRL>
__device__ __forceinline__ float dppf(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void swap16(float &a, float &b)      // a = [a0 b0 a2 b2], b = [a1 b1 a3 b3] (rows of 16 lanes)
{
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}
__device__ __forceinline__ void swap32(float &a, float &b)      // a = [a0 a1 b0 b1], b = [a2 a3 b2 b3]
{
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const float *__restrict__ x, float *__restrict__ out, int passes, int perwave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qbuf = (float *)smem;                                  // [32][64]: a query per half wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int room = (blockIdx.x & 7) >> 1;                       // XCD x (round-robin dispatch) serves room x / 2
    const float *xr = x + (size_t)room * ROOM * C;
    for (int i = tid; i < 32 * C; i += 1024) qbuf[i] = xr[(size_t)((blockIdx.x * 32 + i / C) & (ROOM - 1)) * C + (i % C)];
    __syncthreads();
    const float4 *q4 = (const float4 *)(qbuf + (2 * wave + h) * C);
    float acc_out = 0.f;
    for (int p = 0; p < passes; ++p) {
        const unsigned idx = hash_u32((blockIdx.x * 16u + wave) * 65536u + p * 64u + lane) & (ROOM - 1);
        float4 E[16];
        if (MODE == 0 || MODE == 5) {
            const float4 *pc = (const float4 *)(xr + (size_t)idx * C);
#pragma unroll
            for (int j = 0; j < 16; ++j) E[j] = pc[j];
        } else if (MODE == 1 || MODE == 3) {
            const int g = lane & 3;
            unsigned id4[4];
            id4[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)idx, 0x00, 0xF, 0xF, true);    // quad_perm [0,0,0,0]
            id4[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)idx, 0x55, 0xF, 0xF, true);
            id4[2] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)idx, 0xAA, 0xF, 0xF, true);
            id4[3] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)idx, 0xFF, 0xF, 0xF, true);
            // D[phi][s] -> E[4 s + phi] for now (the register that will hold chunk 4 s + phi of the lane's own finalist)
#pragma unroll
            for (int phi = 0; phi < 4; ++phi) {
                const float4 *pc = (const float4 *)(xr + (size_t)id4[phi] * C) + g;
#pragma unroll
                for (int s = 0; s < 4; ++s) E[4 * s + phi] = pc[4 * s];
            }
            if (MODE == 1) {
                const bool odd = g & 1, up = g & 2;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float *f = (float *)&E[4 * s];            // f[4 phi + comp]
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        // in: lane g holds a[phi] = piece g of finalist phi; out: lane g holds b[p] = piece p of finalist g
                        float a0 = f[c], a1 = f[4 + c], a2 = f[8 + c], a3 = f[12 + c];
                        // stage 1: lanes g ^ 1, register pairs (0, 1) and (2, 3)
                        const float n0 = dppf<0xB1>(a0), n1 = dppf<0xB1>(a1), n2 = dppf<0xB1>(a2), n3 = dppf<0xB1>(a3);   // quad_perm [1,0,3,2]
                        const float b0 = odd ? n1 : a0, b1 = odd ? a1 : n0, b2 = odd ? n3 : a2, b3 = odd ? a3 : n2;
                        // stage 2: lanes g ^ 2, register pairs (0, 2) and (1, 3)
                        const float m0 = dppf<0x4E>(b0), m1 = dppf<0x4E>(b1), m2 = dppf<0x4E>(b2), m3 = dppf<0x4E>(b3);   // quad_perm [2,3,0,1]
                        f[c] = up ? m2 : b0; f[4 + c] = up ? m3 : b1; f[8 + c] = up ? b2 : m0; f[12 + c] = up ? b3 : m1;
                    }
                }
            }
        } else {
            const int rho = lane >> 4, c16 = lane & 15;
            unsigned id4[4];
#pragma unroll
            for (int phi = 0; phi < 4; ++phi) id4[phi] = (unsigned)__shfl((int)idx, c16 + 16 * phi);
#pragma unroll
            for (int phi = 0; phi < 4; ++phi) {
                const float4 *pc = (const float4 *)(xr + (size_t)id4[phi] * C) + rho;
#pragma unroll
                for (int s = 0; s < 4; ++s) E[4 * s + phi] = pc[4 * s];
            }
            if (MODE == 2) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float *f = (float *)&E[4 * s];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        // rows of 16 lanes rho = 0..3 hold piece rho of finalists phi = 0..3 in f[4 phi + c]; wanted: row rho holds pieces
                        // p = 0..3 of finalist rho in f[4 p + c]: a 4 x 4 transpose over (row, register)
                        swap16(f[c], f[4 + c]); swap16(f[8 + c], f[12 + c]);
                        swap32(f[c], f[8 + c]); swap32(f[4 + c], f[12 + c]);
                    }
                }
            }
        }
        if (MODE == 5) {
            float z = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) z += E[j].x;
            acc_out += z;
        } else {
            float z = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 qv = q4[j];
                z = __fmaf_rn(qv.x, E[j].x, z); z = __fmaf_rn(qv.y, E[j].y, z); z = __fmaf_rn(qv.z, E[j].z, z); z = __fmaf_rn(qv.w, E[j].w, z);
            }
            acc_out += z;
        }
    }
    out[(size_t)blockIdx.x * 1024 + tid] = acc_out;
}

template <int MODE>
static float run(const float *dx, float *dout, int blocks, int passes, std::vector<float> *res)
{
    const size_t lds = 150 * 1024;
    hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 1024, lds>>>(dx, dout, passes, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe<MODE><<<blocks, 1024, lds>>>(dx, dout, passes, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (res) { res->resize((size_t)blocks * 1024); hipMemcpy(res->data(), dout, res->size() * 4, hipMemcpyDeviceToHost); }
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) { printf("mode %d: %s\n", MODE, hipGetErrorString(err)); exit(1); }
    return ms / 5;
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, passes = argc > 2 ? atoi(argv[2]) : 48;
    const size_t n = (size_t)ROOMS * ROOM * C;
    std::vector<float> hx(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) * (1.0f / 16777216.0f)) - 0.5f; }
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, (size_t)blocks * 1024 * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> r0, r1, r2;
    const float t0 = run<0>(dx, dout, blocks, passes, &r0);
    const float t1 = run<1>(dx, dout, blocks, passes, &r1);
    const float t2 = run<2>(dx, dout, blocks, passes, &r2);
    const float t3 = run<3>(dx, dout, blocks, passes, nullptr);
    const float t4 = run<4>(dx, dout, blocks, passes, nullptr);
    const float t5 = run<5>(dx, dout, blocks, passes, nullptr);
    size_t bad1 = 0, bad2 = 0;
    for (size_t i = 0; i < r0.size(); ++i) {
        bad1 += memcmp(&r0[i], &r1[i], 4) != 0;
        bad2 += memcmp(&r0[i], &r2[i], 4) != 0;
    }
    const double per = 1e6 / passes;     // ns per pass of a workgroup (16 waves x 64 finalists)
    printf("blocks %d (1 per CU), %d passes of 1024 finalists per workgroup; ns per pass [cycles at 2.4 GHz]\n", blocks, passes);
    const char *names[6] = {"0 lane per row (today)", "1 quad loads + DPP transpose", "2 strided loads + permlane swaps",
                            "3 quad loads, no transpose", "4 strided loads, no transpose", "5 lane per row, no chain"};
    const float t[6] = {t0, t1, t2, t3, t4, t5};
    for (int m = 0; m < 6; ++m) printf("  mode %-36s %8.1f ns  [%6.0f]\n", names[m], t[m] * per, t[m] * per * 2.4);
    printf("bit mismatches against mode 0: mode 1 %zu, mode 2 %zu of %zu\n", bad1, bad2, r0.size());
    return (bad1 || bad2) ? 1 : 0;
}
