// Probe (GPU box): what a selection phase can hide behind on gfx950.
//  (1) do bf16 MFMAs (v_mfma_f32_32x32x16_bf16) of one wave overlap vector work of ANOTHER wave of the same SIMD?
//      (tools/mfma_valu_overlap.hip found that fp32 MFMAs do not: mixed = mean of the pure runs)
//  (2) the same inside ONE wave: independent MFMAs and vector instructions interleaved
//  (3) LDS atomic throughput: ds_add_u32 without / with return on 1024 random bins per row, and on one hot bin
//   hipcc -O3 --offload-arch=gfx950 tools/bf16_overlap_probe.hip -o build/bf16_overlap_probe && build/bf16_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 1000;

// modes: 0 all MFMA (16 per round), 1 all VALU (256 v_fma per round), 2 half / half by wave, 3 same wave interleaved
// (16 MFMA + 256 FMA per round), 4 all integer (512), 5 half MFMA / half integer
__global__ __launch_bounds__(1024) void probe(int mode, float *out)
{
    const int wave = threadIdx.x >> 6;
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float r = 0.f;
    const u32x4 ua = {0x3F803F80u + threadIdx.x, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 A = __builtin_bit_cast(bf16x8, ua), B = A;
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 5) && wave < 8);
    if (mode == 3) {
        f32x16 a0 = {}, a1 = {};
        float c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = x + i;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, a0, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 16; ++j) c[j] = __builtin_fmaf(c[j], y, x);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B, A, a1, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 16; ++j) c[j] = __builtin_fmaf(c[j], y, x);
            }
        }
        r = a0[0] + a1[1];
#pragma unroll
        for (int i = 0; i < 16; ++i) r += c[i];
    } else if (do_mfma) {
        f32x16 a0 = {}, a1 = {};
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B, A, a1, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1];
    } else if (mode >= 4) {
        unsigned c[16];
        const unsigned ux = threadIdx.x * 2654435761u, uy = 0x9E3779B9u;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = ux + i;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = (c[i] ^ uy) + ux;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += (float)c[i];
    } else {
        float c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = x + i;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = __builtin_fmaf(c[i], y, x);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += c[i];
    }
    if (r == 12345.678f) out[0] = r;
}

// LDS atomics: 16 waves, every lane issues 16 atomics per round into its wave's 1024-word region.
// kind 0: ds_add_u32 (no return) random bins; 1: ds_add_rtn_u32 random bins; 2: no return, all lanes one bin; 3: rtn one bin;
// 4: plain ds_write_b32 random; 5: no return, bins spread lane-unique (no conflicts); 6: rtn lane-unique
__global__ __launch_bounds__(1024) void lds_probe(int kind, unsigned *out)
{
    __shared__ unsigned bins[16 * 1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 1024; i += 1024) bins[i] = 0;
    __syncthreads();
    unsigned *my = bins + wave * 1024;
    unsigned s = threadIdx.x * 2654435761u + 12345u, acc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s = s * 1664525u + 1013904223u;
            unsigned b = (kind == 2 || kind == 3) ? 7u : ((kind == 5 || kind == 6) ? (unsigned)(lane + 64 * (j & 15)) : (s >> 22));
            if (kind == 0 || kind == 2 || kind == 5) __hip_atomic_fetch_add(&my[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (kind == 4) my[b] = s;
            else acc += __hip_atomic_fetch_add(&my[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (acc == 0x12345u || bins[threadIdx.x] == 0xFFFFFFFFu) out[0] = acc;
}

int main()
{
    float *out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[6] = {"all 16 waves bf16 MFMA (16 per round)", "all 16 waves FMA (256 per round)", "8 waves MFMA + 8 waves FMA",
                            "same wave: 16 MFMA + 256 FMA interleaved", "all 16 waves integer (512 per round)", "8 waves MFMA + 8 waves integer"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 6; ++mode) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 0, 0, mode, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-44s %.3f ms  (%.0f cycles per round at 2.4 GHz)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / ITER);
        }
    const char *ln[7] = {"ds_add_u32 no-return, random of 1024 bins", "ds_add_rtn_u32, random of 1024 bins", "ds_add_u32 no-return, one bin",
                         "ds_add_rtn_u32, one bin", "ds_write_b32 random", "ds_add_u32 no-return, conflict-free", "ds_add_rtn_u32, conflict-free"};
    for (int rep = 0; rep < 2; ++rep)
        for (int kind = 0; kind < 7; ++kind) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(lds_probe, dim3(256), dim3(1024), 0, 0, kind, (unsigned *)out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // 16 waves x 16 wave-instructions per round per CU
            if (rep) printf("%-44s %.3f ms  (%.1f cycles per wave-instruction per CU at 2.4 GHz)\n", ln[kind], ms, ms * 1e-3 * 2.4e9 / ITER / 256.0);
        }
    return 0;
}
