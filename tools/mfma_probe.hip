// Feasibility probe for a barrier-free MLP chain: ONE wave owns 32 points through L layers of 128 -> 128,
// all four 32-row output tiles of a layer in its own accumulators (64 AGPRs), activations in a wave-private
// LDS region updated in place, weights streamed from L2 through a two-step VGPR ring.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int BLK = 32 * 8 + 8;   // floats per 8-channel block of 32 points
#ifndef PROBE_K8N
#define PROBE_K8N 16
#endif
constexpr int K8N = PROBE_K8N;    // K = 8 * K8N
constexpr int LAYERS = 5;

#define MF(acc, a, b) "v_mfma_f32_32x32x2_f32 %[" #acc "], v" #a ", v" #b ", %[" #acc "]\n\t"
#define MF0(acc, a, b) "v_mfma_f32_32x32x2_f32 %[" #acc "], v" #a ", v" #b ", 0\n\t"
#define STEP_A0(b0, b1, b2, b3)                                                                      \
    MF0(c0, 64, b0) MF0(c1, 68, b0) MF0(c2, 72, b0) MF0(c3, 76, b0) MF(c0, 65, b1) MF(c1, 69, b1) MF(c2, 73, b1) \
    MF(c3, 77, b1) MF(c0, 66, b2) MF(c1, 70, b2) MF(c2, 74, b2) MF(c3, 78, b2) MF(c0, 67, b3) MF(c1, 71, b3) \
    MF(c2, 75, b3) MF(c3, 79, b3)
#define STEP_A(b0, b1, b2, b3)                                                                       \
    MF(c0, 64, b0) MF(c1, 68, b0) MF(c2, 72, b0) MF(c3, 76, b0) MF(c0, 65, b1) MF(c1, 69, b1) MF(c2, 73, b1) \
    MF(c3, 77, b1) MF(c0, 66, b2) MF(c1, 70, b2) MF(c2, 74, b2) MF(c3, 78, b2) MF(c0, 67, b3) MF(c1, 71, b3) \
    MF(c2, 75, b3) MF(c3, 79, b3)
#define STEP_B(b0, b1, b2, b3)                                                                       \
    MF(c0, 80, b0) MF(c1, 84, b0) MF(c2, 88, b0) MF(c3, 92, b0) MF(c0, 81, b1) MF(c1, 85, b1) MF(c2, 89, b1) \
    MF(c3, 93, b1) MF(c0, 82, b2) MF(c1, 86, b2) MF(c2, 90, b2) MF(c3, 94, b2) MF(c0, 83, b3) MF(c1, 87, b3) \
    MF(c2, 91, b3) MF(c3, 95, b3)

// acc[mb] += W[mb] . act   for the four 32-row blocks; w = packed [k8][mb][64 lanes] float4 (+lane)
__device__ __forceinline__ void mac4(const float4 *w, const float *bptr, f32x16 &c0, f32x16 &c1, f32x16 &c2, f32x16 &c3)
{
    unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)bptr;
    const float4 *wp = w;
    int n = K8N / 2 - 2;
    const unsigned long long step = 4096ull;
    constexpr int S1 = BLK * 4, S2 = 2 * BLK * 4;
    asm volatile(
        "global_load_dwordx4 v[64:67], %[wp], off\n\t"
        "global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t"
        "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
        "global_load_dwordx4 v[80:83], %[wp], off\n\t"
        "global_load_dwordx4 v[84:87], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[88:91], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[92:95], %[wp], off offset:3072\n\t"
        "ds_read_b128 v[96:99], %[lds]\n\t"
        "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
        "ds_read_b128 v[100:103], %[lds] offset:%[s1]\n\t"
        "s_waitcnt vmcnt(4) lgkmcnt(1)\n\t"
        STEP_A0(96, 97, 98, 99)
        "global_load_dwordx4 v[64:67], %[wp], off\n\t"
        "global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t"
        "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
        "ds_read_b128 v[96:99], %[lds] offset:%[s2]\n\t"
        "s_waitcnt vmcnt(4) lgkmcnt(1)\n\t"
        STEP_B(100, 101, 102, 103)
        "global_load_dwordx4 v[80:83], %[wp], off\n\t"
        "global_load_dwordx4 v[84:87], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[88:91], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[92:95], %[wp], off offset:3072\n\t"
        "v_add_u32 %[lds], %[s2], %[lds]\n\t"
        "s_cmp_eq_u32 %[n], 0\n\t"
        "s_cbranch_scc1 L_last_%=\n\t"
        "L_loop_%=:\n\t"
        "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
        "ds_read_b128 v[100:103], %[lds] offset:%[s1]\n\t"
        "s_waitcnt vmcnt(4) lgkmcnt(1)\n\t"
        STEP_A(96, 97, 98, 99)
        "global_load_dwordx4 v[64:67], %[wp], off\n\t"
        "global_load_dwordx4 v[68:71], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[72:75], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[76:79], %[wp], off offset:3072\n\t"
        "v_lshl_add_u64 %[wp], %[wp], 0, %[step]\n\t"
        "ds_read_b128 v[96:99], %[lds] offset:%[s2]\n\t"
        "s_waitcnt vmcnt(4) lgkmcnt(1)\n\t"
        STEP_B(100, 101, 102, 103)
        "global_load_dwordx4 v[80:83], %[wp], off\n\t"
        "global_load_dwordx4 v[84:87], %[wp], off offset:1024\n\t"
        "global_load_dwordx4 v[88:91], %[wp], off offset:2048\n\t"
        "global_load_dwordx4 v[92:95], %[wp], off offset:3072\n\t"
        "v_add_u32 %[lds], %[s2], %[lds]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 L_loop_%=\n\t"
        "L_last_%=:\n\t"
        "ds_read_b128 v[100:103], %[lds] offset:%[s1]\n\t"
        "s_waitcnt vmcnt(4) lgkmcnt(1)\n\t"
        STEP_A(96, 97, 98, 99)
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        STEP_B(100, 101, 102, 103)
        "s_nop 15\n\t"
        "s_nop 3\n\t"
        : [c0] "=&a"(c0), [c1] "=&a"(c1), [c2] "=&a"(c2), [c3] "=&a"(c3), [wp] "+v"(wp), [lds] "+v"(lds), [n] "+s"(n)
        : [step] "s"(step), [s1] "n"(S1), [s2] "n"(S2)
        : "memory", "scc", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76",
          "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91",
          "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103");
}

__device__ __forceinline__ void relu_store(float *act, int mb, int j, int h, const f32x16 &v, unsigned &mask)
{
    float *o = act + (size_t)(mb * 4) * BLK + j * 8 + 4 * h;
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        r[i] = fmaxf(v[i], 0.0f);
        mask |= (r[i] != 0.0f ? 1u : 0u) << i;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) *(float4 *)(o + (size_t)g * BLK) = make_float4(r[4 * g], r[4 * g + 1], r[4 * g + 2], r[4 * g + 3]);
}

__global__ __launch_bounds__(64) void chain_kernel(const float4 *w, const float *x, unsigned long long *masks, float *out,
                                                   int tiles_per_wg, int mode)
{
    extern __shared__ float act[];
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < tiles_per_wg; ++t) {
        const size_t tile = (size_t)blockIdx.x * tiles_per_wg + t;
        // "prologue": 32 points x 128 channels from global (row-coalesced), into the k8-block layout
        if (mode & 1) {
#pragma unroll 4
            for (int p = 0; p < 32; p += 2) {
                const int pp = p + h;
                const float4 v = *(const float4 *)(x + ((tile * 32 + pp) & 0xFFFF) * 128 + 4 * j);
                *(float4 *)(act + (size_t)(j >> 1) * BLK + pp * 8 + 4 * (j & 1)) = v;
            }
        }
#pragma unroll 1
        for (int l = 0; l < LAYERS; ++l) {
            f32x16 c0, c1, c2, c3;
            mac4(w + (size_t)l * K8N * 4 * 64 + lane, act + j * 8 + 4 * h, c0, c1, c2, c3);
            unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
            if (mode & 8) {   // no epilogue: keep the accumulators alive with one cheap use
                if (c0[0] + c1[0] + c2[0] + c3[0] == 12345.f) out[lane] = 1.f;
                continue;
            }
            relu_store(act, 0, j, h, c0, m0);
            __builtin_amdgcn_sched_barrier(0);
            relu_store(act, 1, j, h, c1, m1);
            __builtin_amdgcn_sched_barrier(0);
            relu_store(act, 2, j, h, c2, m2);
            __builtin_amdgcn_sched_barrier(0);
            relu_store(act, 3, j, h, c3, m3);
            if (mode & 2)
                masks[((tile * LAYERS + l) & 0xFFFFF) * 64 + lane] = (unsigned long long)(m0 | (m1 << 16)) | ((unsigned long long)(m2 | (m3 << 16)) << 32);
        }
        if (mode & 4) {
#pragma unroll 1
            for (int c = 0; c < 16; ++c) out[((tile * 32 + j) & 0xFFFF) * 32 + c + 16 * h] = act[(size_t)(c >> 3) * BLK + j * 8 + (c & 7)];
        }
    }
    if (lane == 0) {
        masks[(1 << 20) * 64 + blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        masks[(1 << 20) * 64 + blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

int main(int argc, char **argv)
{
    const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 9;
    const int tiles = argc > 2 ? atoi(argv[2]) : 8;
    const size_t wfloats = (size_t)LAYERS * K8N * 4 * 64 * 4;
    std::vector<float> hw(wfloats);
    for (size_t i = 0; i < wfloats; ++i) hw[i] = (float)((i * 2654435761u) % 1000) / 1000.0f * 0.02f - 0.01f;
    float *dw, *dx, *dout;
    unsigned long long *dm;
    (void)hipMalloc(&dw, wfloats * 4);
    (void)hipMemcpy(dw, hw.data(), wfloats * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&dx, (size_t)65536 * 128 * 4);
    (void)hipMemset(dx, 0, (size_t)65536 * 128 * 4);
    (void)hipMalloc(&dm, (size_t)(1 << 20) * 64 * 8 + 65536 * 16);
    (void)hipMalloc(&dout, (size_t)65536 * 32 * 4);
    const size_t lds = (size_t)16 * BLK * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int modes[5] = {8, 0, 1, 3, 7};
    for (int mi = 0; mi < 5; ++mi) {
        const int mode = modes[mi];
        const int nwg = 256 * wg_per_cu;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(chain_kernel, dim3(nwg), dim3(64), lds, 0, (const float4 *)dw, dx, dm, dout, tiles, mode);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)nwg * tiles * LAYERS * 2.0 * 128 * (8.0 * K8N) * 32;
            if (rep == 2) {
                std::vector<unsigned long long> hc(nwg * 2);
                (void)hipMemcpy(hc.data(), dm + (size_t)(1 << 20) * 64, nwg * 16, hipMemcpyDeviceToHost);
                double sc = 0, sr = 0;
                for (int i = 0; i < nwg; ++i) { sc += hc[2 * i]; sr += hc[2 * i + 1]; }
                printf("  clock %.3f GHz  avg wave cycles %.0f  mfma-cycles/wave %.0f  ", sc / sr * 0.1, sc / nwg,
                       (double)tiles * LAYERS * K8N * 16 * 64);
            }
            if (rep == 2)
                printf("mode %d  wg/cu %d tiles %d: %.1f us  %.1f TFLOP/s (%.1f%% of 157.3)\n", mode, wg_per_cu, tiles, ms * 1e3,
                       flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3 * 100);
        }
    }
    hipError_t e = hipGetLastError();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
