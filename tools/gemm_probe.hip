// Probe (GPU box): the fp32 MFMA GEMM of psg_gemm.cuh on the ResGCN shapes, stand-alone, against variants compiled in
// with -DGP_VARIANT=n (see below).  Prints TFLOP/s and the fraction of the 157.3 TFLOP/s fp32 matrix peak.
//   hipcc -O3 --offload-arch=gfx950 -I pointsecguard_amd/csrc -I include tools/gemm_probe.hip -o build/gemm_probe && build/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "psg_gemm.cuh"
using namespace psg;

template <int EPI, int TQ, int TI>
static float run(const GemmArgs &a, int iters)
{
    constexpr int BR = 64 * TQ, BN = 64 * TI;
    dim3 grid((a.rows + BR - 1) / BR, (a.M + BN - 1) / BN);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_rows_kernel<2, 2, EPI, false, TQ, TI>), grid, dim3(256), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_rows_kernel<2, 2, EPI, false, TQ, TI>), grid, dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main()
{
    struct Shape { int rows, K, M, ld_in; const char *name; } shapes[] = {
        {16384, 1792, 1024, 1792, "fusion 16384 x 1792 -> 1024"},
        {16384, 2816, 512, 2816, "prediction.0 (16384 x (1792) -> 512, K as the fused call sees it: 1792)"},
        {16384, 512, 256, 512, "prediction.1 16384 x 512 -> 256"},
        {16384, 1024, 1792, 1024, "fusion backward 16384 x 1024 -> 1792"},
        {4096, 4096, 4096, 4096, "4096^3"},
        // RandLA-Net's coarse levels (8 clouds per launch): 64 x 64 tiles
        {1280, 512, 512, 512, "small: 1280 x 512 -> 512"},
        {5120, 256, 256, 256, "small: 5120 x 256 -> 256"},
        {20480, 128, 128, 128, "small: 20480 x 128 -> 128"},
        {640, 1024, 1024, 1024, "small: 640 x 1024 -> 1024"},
        {16384, 64, 128, 64, "small: per-vertex 16384 x 64 -> 128 (ResGCN)"},
    };
    for (auto &s : shapes) {
        const int K = s.K == 2816 ? 1792 : s.K;
        float *in, *w, *out, *sc, *sh; uint32_t *mask;
        hipMalloc(&in, (size_t)s.rows * s.ld_in * 4); hipMalloc(&w, (size_t)s.M * K * 4); hipMalloc(&out, (size_t)s.rows * s.M * 4);
        hipMalloc(&sc, s.M * 4); hipMalloc(&sh, s.M * 4); hipMalloc(&mask, (size_t)s.rows * ((s.M + 31) / 32) * 4);
        std::vector<float> h(std::max((size_t)s.rows * s.ld_in, (size_t)s.M * K));
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(w, h.data(), (size_t)s.M * K * 4, hipMemcpyHostToDevice);
        hipMemcpy(sc, h.data(), s.M * 4, hipMemcpyHostToDevice); hipMemcpy(sh, h.data(), s.M * 4, hipMemcpyHostToDevice);
        GemmArgs a;
        a.in = in; a.w = w; a.bias = sh; a.gbias = nullptr; a.scale = sc; a.shift = sh; a.sq = nullptr; a.out = out; a.mask_out = mask; a.mask_in = nullptr;
        a.rows = s.rows; a.K = K; a.M = s.M; a.ld_in = s.ld_in; a.ld_w = K; a.ld_out = s.M; a.group_rows = 1; a.accumulate = 0; a.addend = nullptr; a.ld_add = 0;
        const double fl = 2.0 * s.rows * (double)K * s.M;
        const bool small = s.name[0] == 's';
        const float t_relu = small ? run<EPI_RELU_AFFINE, 1, 1>(a, 50) : run<EPI_RELU_AFFINE, 2, 2>(a, 10);
        a.bias = nullptr; a.scale = nullptr; a.shift = nullptr; a.mask_out = nullptr;
        const float t_lin = small ? run<EPI_LINEAR, 1, 1>(a, 50) : run<EPI_LINEAR, 2, 2>(a, 10);
        printf("%-72s relu+affine %.1f us %.1f TF (%.2f)   linear %.1f us %.1f TF (%.2f)\n", s.name, t_relu * 1e3, fl / t_relu * 1e-9, fl / t_relu * 1e-9 / 157.3,
               t_lin * 1e3, fl / t_lin * 1e-9, fl / t_lin * 1e-9 / 157.3);
        hipFree(in); hipFree(w); hipFree(out); hipFree(sc); hipFree(sh); hipFree(mask);
    }
    return 0;
}
