// Probe (GPU box): is a chain of fp32 MFMAs bit-identical to the ascending-k fmaf chain of a CPU sgemm?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_chain_probe.hip -o /tmp/mfma_chain_probe && /tmp/mfma_chain_probe
// For random A[32][64], B[32][64] (and a badly conditioned set) it compares dot(A_i, B_j) from
//   (1) v_mfma_f32_32x32x2_f32, 32 chained instructions (k = 2 s + (lane >> 5)),
//   (2) v_mfma_f32_16x16x4_f32, 16 chained instructions (k = 4 s + (lane >> 4)),
// with the host's fmaf chain  dot = a0*b0; dot = fmaf(a_k, b_k, dot), and with two other orders of the four products
// inside one 16x16x4 instruction to see which one the hardware implements.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void probe(const float *A, const float *B, float *out32, float *out16)
{
    const int lane = threadIdx.x;
    {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int j = lane & 31, h = lane >> 5;
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j * 64 + 2 * s + h], B[j * 64 + 2 * s + h], acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) out32[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];   // [row of A][row of B]
    }
    {
        const int q = lane & 15, g = lane >> 4;
        for (int ta = 0; ta < 2; ++ta)
            for (int tb = 0; tb < 2; ++tb) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int s = 0; s < 16; ++s)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * ta + q) * 64 + 4 * s + g], B[(16 * tb + q) * 64 + 4 * s + g], acc, 0, 0, 0);
                for (int r = 0; r < 4; ++r) out16[(16 * ta + 4 * g + r) * 32 + 16 * tb + q] = acc[r];
            }
    }
}

static float chain(const float *a, const float *b)
{
    float d = a[0] * b[0];
    for (int k = 1; k < 64; ++k) d = fmaf(a[k], b[k], d);
    return d;
}
static float chain_pairs4(const float *a, const float *b)   // ((p0 + p1) + (p2 + p3)) added to the accumulator per group of 4
{
    float d = 0.f;
    for (int s = 0; s < 16; ++s) {
        const float *x = a + 4 * s, *y = b + 4 * s;
        d = d + ((x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]));
    }
    return d;
}
static float chain_rev4(const float *a, const float *b)     // fmaf chain, descending k inside each group of 4
{
    float d = 0.f;
    for (int s = 0; s < 16; ++s)
        for (int t = 3; t >= 0; --t) d = fmaf(a[4 * s + t], b[4 * s + t], d);
    return d;
}

int main()
{
    std::vector<float> A(32 * 64), B(32 * 64), o32(32 * 32), o16(32 * 32);
    float *dA, *dB, *d32, *d16;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d32, o32.size() * 4); hipMalloc(&d16, o16.size() * 4);
    srand(1);
    long n = 0, bad32 = 0, bad16 = 0, p4 = 0, r4 = 0;
    for (int trial = 0; trial < 400; ++trial) {
        const float scale = trial % 2 ? 20.f : 1.f;
        for (auto &v : A) v = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f) + (trial % 4 == 3 ? 5.f : 0.f);
        for (auto &v : B) v = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, d32, d16);
        hipMemcpy(o32.data(), d32, o32.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(o16.data(), d16, o16.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float ref = chain(&A[i * 64], &B[j * 64]);
                ++n;
                bad32 += memcmp(&ref, &o32[i * 32 + j], 4) != 0;
                bad16 += memcmp(&ref, &o16[i * 32 + j], 4) != 0;
                const float a = chain_pairs4(&A[i * 64], &B[j * 64]), b = chain_rev4(&A[i * 64], &B[j * 64]);
                p4 += memcmp(&a, &o16[i * 32 + j], 4) != 0;
                r4 += memcmp(&b, &o16[i * 32 + j], 4) != 0;
            }
    }
    printf("dots compared %ld: 32x32x2 chain != fmaf chain: %ld ; 16x16x4 chain != fmaf chain: %ld ; (16x16x4 vs pairwise-in-4: %ld, vs "
           "descending-in-4: %ld mismatches)\n", n, bad32, bad16, p4, r4);
    return 0;
}
