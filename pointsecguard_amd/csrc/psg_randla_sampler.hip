// Possibility-based crop sampler of the RandLA-Net input pipeline on the device ("next").
//
// Reference: RandLA-Net/main_S3DIS.py:116-187 (get_batch_gen / spatially_regular_gen).  Per crop the reference picks the
// point with the lowest "possibility" of the cloud with the lowest minimum, jitters it, asks the cloud's sklearn KDTree for
// the num_points = 40 960 nearest points (float64 squared distances, sorted ascending), shuffles them, and raises the
// possibility of the points it took by (1 - d / max d)^2 (d in float32).  The tree query of 40 960 neighbours out of a few
// hundred thousand points is the expensive part; on the device it is a distance pass + one radix sort.
//
// Device side (one sampler object per cloud): points [M][3] f32 (the tree holds their exact float64 promotion; the pick
// point - a tree point plus float64 noise - is float64 and is passed as such) and possibility [M] f64 resident; argmin = two atomic
// passes (the value, then the lowest index that holds it: np.argmin's first-index rule); query = float64 squared
// distances in the tree's arithmetic (data and query promoted to double, (dx^2 + dy^2) + dz^2), their bit patterns as
// sort keys (non-negative doubles order like their bits), stable hipcub radix sort -> ascending (distance, index), the
// first k indices; update = the reference's float32 expression, added to the float64 possibility.  The host keeps what the
// reference draws from numpy's global generator (initial possibilities, jitter, shuffle, duplication), in its order.
#include <hipcub/hipcub.hpp>

#include "psg_common.h"

using namespace psg;

struct psg_rla_sampler {
    psg_ctx *ctx;
    int M;
    float *pts;                       // [M][3]
    double *poss;                     // [M]
    unsigned long long *keys, *keys_sorted;
    int32_t *iota, *order;            // order = indices ascending by (distance, index)
    void *tmp;
    size_t tmp_bytes;
    unsigned long long *min_bits;     // [1] bit pattern of the smallest possibility
    int32_t *min_idx;                 // [1]
    float *dmax;                      // [1] largest float32 distance of the last update (as ordered bits)
};

namespace {

__global__ void iota_kernel(int32_t *v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

__global__ void dist_keys_kernel(const float *__restrict__ pts, int n, double px, double py, double pz, unsigned long long *__restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dx = (double)pts[3 * i] - px, dy = (double)pts[3 * i + 1] - py, dz = (double)pts[3 * i + 2] - pz;
    const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
    keys[i] = (unsigned long long)__double_as_longlong(d);
}

__global__ void min_bits_kernel(const double *__restrict__ p, int n, unsigned long long *__restrict__ out)
{
    unsigned long long m = ~0ull;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(p[i]);      // possibilities are >= 0
        m = b < m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = ((unsigned long long)__shfl_xor((unsigned)(m >> 32), o) << 32) | __shfl_xor((unsigned)m, o);
        m = t < m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMin(out, m);
}

__global__ void min_index_kernel(const double *__restrict__ p, int n, const unsigned long long *__restrict__ bits, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (unsigned long long)__double_as_longlong(p[i]) == bits[0]) atomicMin(out, i);
}

// float32 distances of the taken points in the reference's arithmetic (main_S3DIS.py:163): the points are the KDTree's
// float64 copy of the cloud and the pick point is float64, so the DIFFERENCE is taken in float64 and then rounded to
// float32 (`(points[idx] - pick_point).astype(np.float32)`), squared and summed ((dx^2 + dy^2) + dz^2) in float32; their
// maximum through ordered bits (non-negative floats)
__global__ void upd_max_kernel(const float *__restrict__ pts, const int32_t *__restrict__ idx, int k, double px, double py, double pz,
                               float *__restrict__ d_out, unsigned *__restrict__ dmax_bits)
{
    float mx = 0.0f;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x) {
        const int i = idx[j];
        const float dx = (float)__dsub_rn((double)pts[3 * i], px), dy = (float)__dsub_rn((double)pts[3 * i + 1], py),
                    dz = (float)__dsub_rn((double)pts[3 * i + 2], pz);
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        d_out[j] = d;
        mx = fmaxf(mx, d);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) atomicMax(dmax_bits, __float_as_uint(mx));
}

__global__ void upd_apply_kernel(const int32_t *__restrict__ idx, const float *__restrict__ d, int k, const unsigned *__restrict__ dmax_bits,
                                 double *__restrict__ poss)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const float t = __fsub_rn(1.0f, __fdiv_rn(d[j], __uint_as_float(dmax_bits[0])));
    poss[idx[j]] += (double)__fmul_rn(t, t);      // (the taken indices are distinct)
}

}  // namespace

extern "C" int psg_rla_sampler_destroy(psg_rla_sampler *s);

extern "C" int psg_rla_sampler_create(psg_ctx *ctx, const float *points_host, const double *possibility_host, int n_points,
                                      psg_rla_sampler **out)
{
    PSG_REQUIRE(ctx && points_host && possibility_host && out && n_points > 0, "psg_rla_sampler_create: bad argument");
    PSG_CHECK_HIP(hipSetDevice(ctx->device));
    psg_rla_sampler *s = new psg_rla_sampler();
    s->ctx = ctx; s->M = n_points;
    const size_t M = (size_t)n_points;
    size_t tb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                             (int32_t *)nullptr, (int32_t *)nullptr, n_points, 0, 64, (hipStream_t)0);
    s->tmp_bytes = tb;
    void **slots[] = {(void **)&s->pts, (void **)&s->poss, (void **)&s->keys, (void **)&s->keys_sorted, (void **)&s->iota,
                      (void **)&s->order, &s->tmp, (void **)&s->min_bits, (void **)&s->min_idx, (void **)&s->dmax};
    const size_t sizes[] = {M * 12, M * 8, M * 8, M * 8, M * 4, M * 4, tb ? tb : 16, 8, 4, 4};
    for (int i = 0; i < 10; ++i) {
        *slots[i] = nullptr;
        if (hipMalloc(slots[i], sizes[i]) != hipSuccess) {
            set_error("psg_rla_sampler_create: hipMalloc(%zu) failed", sizes[i]);
            for (int j = 0; j < i; ++j) (void)hipFree(*slots[j]);
            delete s;
            return PSG_ERR_HIP;
        }
    }
    // (from here on a failure releases the sampler and its buffers)
    hipError_t e = psg::copy_sync(s->pts, points_host, M * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = psg::copy_sync(s->poss, possibility_host, M * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        // (a stream of its own, not stream 0 + hipDeviceSynchronize: a device synchronise invalidates any hipGraph capture
        // another host thread has in progress - psg_common.h)
        hipStream_t cst = nullptr;
        e = hipStreamCreateWithFlags(&cst, hipStreamNonBlocking);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(iota_kernel, dim3((n_points + 255) / 256), dim3(256), 0, cst, s->iota, n_points);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(cst);
            (void)hipStreamDestroy(cst);
        }
    }
    if (e != hipSuccess) {
        set_error("psg_rla_sampler_create: %s", hipGetErrorString(e));
        (void)psg_rla_sampler_destroy(s);
        return PSG_ERR_HIP;
    }
    *out = s;
    return PSG_OK;
}

extern "C" int psg_rla_sampler_destroy(psg_rla_sampler *s)
{
    if (!s) return PSG_OK;
    void *all[] = {s->pts, s->poss, s->keys, s->keys_sorted, s->iota, s->order, s->tmp, s->min_bits, s->min_idx, s->dmax};
    for (void *p : all) (void)hipFree(p);
    delete s;
    return PSG_OK;
}

// np.argmin / np.min of the possibility (first index on equal values); synchronises the stream
extern "C" int psg_rla_sampler_argmin(psg_rla_sampler *s, int *index_out, double *value_out, psg_stream stream)
{
    PSG_REQUIRE(s && index_out && value_out, "psg_rla_sampler_argmin: null argument");
    PSG_CHECK_HIP(hipSetDevice(s->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    PSG_CHECK_HIP(hipMemsetAsync(s->min_bits, 0xFF, 8, st));
    PSG_CHECK_HIP(hipMemsetAsync(s->min_idx, 0x7F, 4, st));
    hipLaunchKernelGGL(min_bits_kernel, dim3(256), dim3(256), 0, st, s->poss, s->M, s->min_bits);
    PSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(min_index_kernel, dim3((s->M + 255) / 256), dim3(256), 0, st, s->poss, s->M, s->min_bits, s->min_idx);
    PSG_LAUNCH_CHECK();
    unsigned long long bits = 0;
    int32_t idx = 0;
    PSG_CHECK_HIP(hipMemcpyAsync(&bits, s->min_bits, 8, hipMemcpyDeviceToHost, st));
    PSG_CHECK_HIP(hipMemcpyAsync(&idx, s->min_idx, 4, hipMemcpyDeviceToHost, st));
    PSG_CHECK_HIP(hipStreamSynchronize(st));
    *index_out = idx;
    memcpy(value_out, &bits, 8);
    return PSG_OK;
}

// the k points nearest to `pick` (KDTree.query(pick, k)[1][0]): out_idx [k] device, ascending (float64 distance, index)
extern "C" int psg_rla_sampler_query(psg_rla_sampler *s, const double *pick_host3, int k, int32_t *out_idx, psg_stream stream)
{
    PSG_REQUIRE(s && pick_host3 && out_idx && k > 0 && k <= s->M, "psg_rla_sampler_query: k=%d outside [1, %d]", k, s ? s->M : 0);
    PSG_CHECK_HIP(hipSetDevice(s->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dist_keys_kernel, dim3((s->M + 255) / 256), dim3(256), 0, st, s->pts, s->M, pick_host3[0], pick_host3[1],
                       pick_host3[2], s->keys);
    PSG_LAUNCH_CHECK();
    size_t tb = s->tmp_bytes;
    PSG_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(s->tmp, tb, s->keys, s->keys_sorted, s->iota, s->order, s->M, 0, 64, st));
    PSG_CHECK_HIP(hipMemcpyAsync(out_idx, s->order, (size_t)k * 4, hipMemcpyDeviceToDevice, st));
    return PSG_OK;
}

// possibility[idx] += (1 - d / max d)^2 for the k taken points (main_S3DIS.py:163-165); idx [k] device, scratch [k] floats
extern "C" int psg_rla_sampler_update(psg_rla_sampler *s, const int32_t *idx, int k, const double *pick_host3, float *scratch,
                                      psg_stream stream)
{
    PSG_REQUIRE(s && idx && pick_host3 && scratch && k > 0, "psg_rla_sampler_update: bad argument");
    PSG_CHECK_HIP(hipSetDevice(s->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    PSG_CHECK_HIP(hipMemsetAsync(s->dmax, 0, 4, st));
    hipLaunchKernelGGL(upd_max_kernel, dim3(128), dim3(256), 0, st, s->pts, idx, k, pick_host3[0], pick_host3[1], pick_host3[2], scratch,
                       (unsigned *)s->dmax);
    PSG_LAUNCH_CHECK();
    hipLaunchKernelGGL(upd_apply_kernel, dim3((k + 255) / 256), dim3(256), 0, st, idx, scratch, k, (const unsigned *)s->dmax, s->poss);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// test read-back: the possibility array [M] to the host (synchronises)
extern "C" int psg_rla_sampler_possibility(psg_rla_sampler *s, double *host_out)
{
    PSG_REQUIRE(s && host_out, "psg_rla_sampler_possibility: null argument");
    PSG_CHECK_HIP(hipDeviceSynchronize());
    PSG_CHECK_HIP(psg::copy_sync(host_out, s->poss, (size_t)s->M * 8, hipMemcpyDeviceToHost));
    return PSG_OK;
}
