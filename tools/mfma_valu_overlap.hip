// Probe (GPU box): can a SIMD of gfx950 run fp32 MFMAs of one wave and fp32 vector FMAs of another wave at the same time?
// 16 waves per workgroup, one workgroup per CU (4 waves per SIMD).  Every wave runs ITER rounds of either 32
// v_mfma_f32_16x16x4_f32 (two chains) or 256 v_fma_f32 (16 chains).  Timed: all waves MFMA, all waves FMA, and waves
// alternating by SIMD slot (waves 0-7 MFMA, 8-15 FMA: every SIMD holds two of each).  If the two pipes overlap, the mixed
// run takes about max(half, half) of the pure runs; if they share the issue slot / datapath, about their mean.
// Modes 3 / 4 repeat it with integer vector work (v_xor + v_add) instead of FMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o build/mfma_valu_overlap && build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int ITER = 2000;

__global__ __launch_bounds__(1024) void probe(int mode, float *out)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 4) && wave < 8);
    const bool integer = mode >= 3;
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float r = 0.f;
    if (do_mfma) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1];
    } else if (integer) {
        unsigned c[16];
        const unsigned ux = threadIdx.x * 2654435761u, uy = 0x9E3779B9u;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = ux + i;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = (c[i] ^ uy) + ux;        // v_xor + v_add: 512 integer ops per round
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

int main()
{
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[5] = {"all 16 waves MFMA (32 per round)", "all 16 waves FMA (256 per round)", "8 waves MFMA + 8 waves FMA",
                            "all 16 waves integer (512 per round)", "8 waves MFMA + 8 waves integer"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 0, 0, mode, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-40s %.3f ms  (%.0f cycles per round at 2.4 GHz)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / ITER);
        }
    return 0;
}
