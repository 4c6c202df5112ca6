// Fused feature-space kNN kernels of the ResGCN path (C = 64) in their own translation unit: compiled with the MFMA
// accumulators in VGPRs (every accumulator is read by vector compares right after the matrix instructions: the AGPR form
// costs one v_accvgpr_read per value) and with -ffp-contract=off (the exact distances follow the reference's fp32
// evaluation order, SURVEY 8a').  psg_resgcn.hip calls the launchers below (psg_knn.h).
#include "psg_knn.h"
#include "psg_common.h"

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "psg_wave.cuh"
#include "psg_knn_ops.cuh"

namespace {
#include "psg_knn_fused.cuh"
#include "psg_knn_bf.cuh"

// xp / bp / sq for the stand-alone entry point: one wave per point, lane = feature (C = 64)
__global__ void knn_prep_kernel(const float *__restrict__ x, int ld, size_t rows, float *__restrict__ xp, float *__restrict__ sq,
                                unsigned short *__restrict__ bp)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t v = t >> 6;
    if (v >= rows) return;
    const int c = (int)(t & 63);
    const float val = x[v * ld + c];
    knn_store_xp(xp, v, c, val);
    if (bp) knn_store_bp(bp, v, c, val);
    const float s = knn_wave_sumsq(val, threadIdx.x & 63);
    if (c == 0) {
        sq[v] = s;
        if (bp) knn_store_aug((uint4 *)bp, v, s);
    }
}

// The prefilter kernel's sample cut (psg_knn_bf.cuh): step 0 of a workgroup sees S = 32 * nv0 of the room's N candidates;
// the row's threshold is the r-th smallest sample key (+ margin).  The row ends short of KK keys below it iff the sample
// holds at least r of the row's KK - 1 nearest candidates: X ~ Hypergeometric(N, KK - 1, S) for candidates in random
// order.  r = the smallest value with P(X >= r) <= 1e-7 (S = N: r = KK, the plain exact cut).
unsigned knn_sample_rank_S(int N, int KK, int S);
unsigned knn_sample_rank(int N, int KK, int nsamp)
{
    const int tpr = N >> 5, n_steps = (tpr + KB_WAVES - 1) / KB_WAVES, nv0 = (tpr + n_steps - 1) / n_steps;
    return knn_sample_rank_S(N, KK, 32 * nv0 * nsamp);
}
unsigned knn_sample_rank_S(int N, int KK, int S)
{
    static std::mutex mu;
    static std::unordered_map<unsigned long long, unsigned> cache;
    const unsigned long long key = ((unsigned long long)(unsigned)N << 32) | ((unsigned long long)(unsigned)KK << 16) | (unsigned)S;
    std::lock_guard<std::mutex> lock(mu);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int K1 = KK - 1;
    S = S < N ? S : N;
    auto lchoose = [](int n, int m) { return std::lgamma((double)n + 1.0) - std::lgamma((double)m + 1.0) - std::lgamma((double)(n - m) + 1.0); };
    const double denom = lchoose(N, S);
    const int hi = K1 < S ? K1 : S;
    int r = hi + 1;
    double tail = 0.0;
    for (int x = hi; x >= 0; --x) {
        const double p = (S - x > N - K1) ? 0.0 : std::exp(lchoose(K1, x) + lchoose(N - K1, S - x) - denom);
        if (tail + p > 1e-7) break;
        tail += p;
        r = x;
    }
    r = r < 1 ? 1 : (r > S ? S : r);
    cache[key] = (unsigned)r;
    return (unsigned)r;
}
}  // namespace

namespace psg {

size_t knn_xp_bytes(size_t rows) { return rows * 64 * sizeof(float); }
size_t knn_bp_bytes(size_t rows) { return (rows + 31) / 32 * 9 * 64 * 16; }

hipError_t knn_setup()
{
    static std::once_flag once;
    static hipError_t rc = hipSuccess;
    std::call_once(once, [] {
        rc = hipFuncSetAttribute((const void *)knn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_fused_lds_bytes());
        if (rc == hipSuccess)
            rc = hipFuncSetAttribute((const void *)knn_bf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)knn_bf_lds_bytes());
    });
    return rc;
}

bool knn_shape_ok(int N, int k, int d, KnnPath path)
{
    if (N <= 0 || N > 4096 || k > 16 || (k - 1) * d + 1 > 448) return false;
    return path == KNN_PATH_BF16 ? (N % 32) == 0 : (N % 16) == 0;
}

hipError_t knn_prep_launch(const float *x, int ld, size_t rows, const KnnBuffers &buf, bool want_bp, hipStream_t st)
{
    hipLaunchKernelGGL(knn_prep_kernel, dim3((unsigned)((rows * 64 + 255) / 256)), dim3(256), 0, st, x, ld, rows, buf.xp, buf.sq,
                       want_bp ? (unsigned short *)buf.bp : (unsigned short *)nullptr);
    return hipGetLastError();
}

hipError_t knn_launch(const KnnBuffers &buf, int B, int N, int k, int d, int32_t *out, KnnPath path, hipStream_t st)
{
    KnnFusedArgs f;
    f.xp = buf.xp; f.sq = buf.sq; f.out = out; f.N = N; f.k = k; f.d = d;
    f.KK = (k - 1) * d + 1;
    f.magic = (unsigned)(((1u << 18) + (unsigned)d - 1u) / (unsigned)d);
    const int slack = KF_CAP - f.KK;
    f.TOL = slack / 16;
    f.LOW = f.KK + slack / 4;
    static const int xcd_order = psg::env_int("PSG_KNN_XCD_ORDER", 1);
    f.xcd_order = xcd_order;
    f.rsel = 0u;
    if (path == KNN_PATH_F32) {
        // one sampled cut instead of the growth schedule where that saves cuts (KK >= PSG_KNN_F32_SAMPLED_KK): the first CAP
        // candidates of a row, spread over the room, give the threshold (psg_knn_fused.cuh: KnnFusedArgs::rsel)
        static const int sampled_from = psg::env_int("PSG_KNN_F32_SAMPLED_KK", 150);
        if (f.KK >= sampled_from && N > KF_CAP) f.rsel = knn_sample_rank_S(N, f.KK, KF_CAP);
        hipLaunchKernelGGL(knn_fused_kernel, dim3((unsigned)((size_t)B * N / KF_Q)), dim3(KF_WAVES * 64), knn_fused_lds_bytes(), st, f);
        return hipGetLastError();
    }
    if (!buf.x || buf.ld < 64 || (buf.ld & 3)) return hipErrorInvalidValue;      // the prefilter path reads its finalists from the row-major features
    KnnBfArgs a;
    a.bp = (const kb_u32x4 *)buf.bp; a.sq = buf.sq; a.out = out; a.N = N; a.k = k; a.d = d; a.KK = f.KK; a.magic = f.magic;
    // two sample tiles per wave for the longest rows (psg_knn_bf.cuh).  Round 4 switched them on from KK = 311 (d >= 21): rows then
    // end with 1.45 KK entries instead of 1.7 KK, which kept them inside their 1024-entry buffers AND under the 256-finalist
    // limit.  Since round 6 (2048-bin final ranking: 64 finalists per row at d = 27) only the buffers matter, and on the
    // network's features 1.7 KK = 570 - 690 entries fit: one sample tile is 10 - 12 us faster per call at d = 21..27 (the
    // second tile's cut costs more than the shorter rows save; tools/r06_f.sh: 158 / 154 / 157 / 154 -> 147 / 142 / 144 / 144 us,
    // one exact-path tile in 165 888 over a 12-iteration attack either way), so the second tile starts at KK = 430 now
    static const int two_from = psg::env_int("PSG_KNN_SAMPLE2_KK", 430);
    const int n_steps = ((N >> 5) + KB_WAVES - 1) / KB_WAVES;
    a.nsamp = (a.KK >= two_from && (N & 511) == 0 && n_steps >= 4) ? 2 : 1;
    a.rsel = knn_sample_rank(N, a.KK, a.nsamp);
    a.x = buf.x; a.ld = buf.ld;
    static const int fine_from = psg::env_int("PSG_KNN_FINE_CUT_KK", 100);
    a.fine_cut = a.KK >= fine_from;
    a.exact = f;
    a.stats = buf.stats;
    hipLaunchKernelGGL(knn_bf_kernel, dim3((unsigned)((size_t)B * N / KB_Q)), dim3(KB_WAVES * 64), knn_bf_lds_bytes(), st, a);
    return hipGetLastError();
}

}  // namespace psg

#ifdef PSG_KF_STAMP
// diagnostic build only: read and clear the fused kNN kernel's phase stamps (tools/knn_stamp.py)
extern "C" int psg_dbg_knn_stamps(unsigned long long *host_out)
{
    PSG_CHECK_HIP(hipDeviceSynchronize());
    PSG_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_kf_stamps), 16 * sizeof(unsigned long long)));
    unsigned long long zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    PSG_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_kf_stamps), zero, sizeof(zero)));
    return PSG_OK;
}
#endif
#ifdef PSG_KF_TL
extern "C" int psg_dbg_knn_ph(unsigned long long *host_out)      // [32], see g_kb_ph
{
    PSG_CHECK_HIP(hipDeviceSynchronize());
    PSG_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_kb_ph), 32 * sizeof(unsigned long long)));
    static unsigned long long zero[32];
    PSG_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_kb_ph), zero, sizeof(zero)));
    return PSG_OK;
}
extern "C" int psg_dbg_knn_tl(unsigned long long *host_out)      // [32], see g_kf_tl
{
    PSG_CHECK_HIP(hipDeviceSynchronize());
    PSG_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_kf_tl), 32 * sizeof(unsigned long long)));
    static unsigned long long zero[32];
    PSG_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_kf_tl), zero, sizeof(zero)));
    return PSG_OK;
}
#endif

