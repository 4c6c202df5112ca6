#include <mutex>
// Context, error reporting and version of libpsg.so.
#include <atomic>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "psg_common.h"

namespace {
thread_local char g_err[512] = "";
}

namespace psg {
// The switches libpsg knows, with what they are allowed to change: 'p' = selects another TESTED path (same results, to
// rounding where the path sums in another order), 'd' = diagnosis output only, 'r' = changes results (timing bisection:
// present in -DPSG_DIAG_BUILD libraries only, so the default library lists none).
struct EnvSwitch { const char *name; char kind; };
static const EnvSwitch kEnvSwitches[] = {
    {"PSG_TRACE_SYNC", 'd'},       {"PSG_GCN_KNN_STATS", 'd'},   {"PSG_KNN_XCD_ORDER", 'p'},  {"PSG_KNN_FINE_CUT_KK", 'p'},  {"PSG_KNN_SAMPLE2_KK", 'p'},  {"PSG_KNN_F32_SAMPLED_KK", 'p'},
    {"PSG_FP1_WAVE", 'p'},         {"PSG_RLA_NO_FUSE16", 'p'},   {"PSG_RLA_ATOMICS", 'p'},    {"PSG_RLA_NO_SPLIT", 'p'},
    {"PSG_RLA_NO_GRAPH", 'p'},     {"PSG_GEMM_SMALL_BELOW", 'p'}, {"PSG_GCN_KNN", 'p'},        {"PSG_GCN_KNN_BF_MAXD", 'p'},
    {"PSG_GCN_PQ_FUSION", 'p'},    {"PSG_GCN_NO_GRAPH", 'p'},    {"PSG_GCN_EDGE_BWD", 'p'},   {"PSG_BALL_QUERY", 'p'},
    {"PSG_PN2_SPLIT", 'p'},          {"PSG_RLA_NO_DIRECT", 'p'},   {"PSG_NU_NO_GRAPH", 'p'},      {"PSG_PN2_GRAPH", 'p'},
    {"PSG_PN2_FPSPLIT", 'p'},        {"PSG_FPS_CFG", 'p'},         {"PSG_PN2_L1T_COLOUR", 'p'},
    {"PSG_PN2_PGD_FUSE", 'p'},
#ifdef PSG_DIAG_BUILD
    {"PSG_DIAG", 'r'},
#endif
};

hipError_t copy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (bytes == 0) return hipSuccess;
    // (a stream of its own per call, not one kept for the process: these are creation-time copies, and a stream that stays
    // alive takes part in the runtime's stream -> hardware-queue assignment of everything created after it)
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dst, src, bytes, kind, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    return e;
}

static std::atomic<uint64_t> g_generation{0};
uint64_t next_generation() { return ++g_generation; }

static std::atomic<long long> g_capture[4];
void capture_note(CaptureCounters *own, int tried, int failed, int replays, int eager)
{
    if (own) { own->tried += tried; own->failed += failed; own->replays += replays; own->eager += eager; }
    g_capture[0] += tried; g_capture[1] += failed; g_capture[2] += replays; g_capture[3] += eager;
}

hipError_t memset_sync(void *dst, int value, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(dst, value, bytes, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    return e;
}

const char *env_str(const char *name)
{
    bool known = false;
    for (const EnvSwitch &e : kEnvSwitches) known = known || strcmp(e.name, name) == 0;
    if (!known) fprintf(stderr, "[psg] environment switch %s is read but not registered in psg_api.hip\n", name);
    return getenv(name);
}

int env_int(const char *name, int dflt)
{
    const char *v = env_str(name);
    return v ? atoi(v) : dflt;
}

bool trace_sync_enabled()
{
    static const bool on = env_int("PSG_TRACE_SYNC", 0) != 0;
    return on;
}

void trace_sync_point(const char *file, int line)
{
    static unsigned long n = 0;
    const char *base = strrchr(file, '/');
    fprintf(stderr, "[psg trace] launch %lu at %s:%d issued\n", ++n, base ? base + 1 : file, line);
    fflush(stderr);
    hipError_t e = hipDeviceSynchronize();
    fprintf(stderr, "[psg trace] launch %lu done (%s)\n", n, hipGetErrorString(e));
    fflush(stderr);
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace psg

extern "C" const char *psg_last_error(void) { return g_err; }

// "NAME=value:kind;..." for every registered switch that is set in this process's environment ("" = all defaults).
extern "C" int psg_env_switches(char *buf, int cap)
{
    PSG_REQUIRE(buf && cap > 0, "psg_env_switches: null / empty buffer");
    std::string out;
    int n = 0;
    for (const psg::EnvSwitch &e : psg::kEnvSwitches) {
        const char *v = getenv(e.name);
        if (!v) continue;
        out += std::string(e.name) + "=" + v + ":" + e.kind + ";";
        ++n;
    }
    PSG_REQUIRE((int)out.size() < cap, "psg_env_switches: buffer of %d bytes is too small (%d needed)", cap, (int)out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return n;
}

// 1 when this library was built with -DPSG_DIAG_BUILD (work-skipping timing switches compiled in), else 0.
extern "C" int psg_diag_build(void)
{
#ifdef PSG_DIAG_BUILD
    return 1;
#else
    return 0;
#endif
}

// Process-wide hipGraph bookkeeping: out[4] = {captures tried, captures failed, graph replays, windows / iterations that ran
// eagerly although a graph was wanted}.
extern "C" int psg_capture_stats(long long *out4)
{
    PSG_REQUIRE(out4, "psg_capture_stats: null out pointer");
    for (int i = 0; i < 4; ++i) out4[i] = psg::g_capture[i].load();
    return PSG_OK;
}

extern "C" const char *psg_version(void) { return "libpsg 0.1 (gfx950, fp32 MFMA)"; }

extern "C" int psg_ctx_create(int device, psg_ctx **out)
{
    PSG_REQUIRE(out, "psg_ctx_create: null out pointer");
    int n = 0;
    PSG_CHECK_HIP(hipGetDeviceCount(&n));
    PSG_REQUIRE(device >= 0 && device < n, "psg_ctx_create: device %d not present (%d devices)", device, n);
    hipDeviceProp_t prop;
    PSG_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        psg::set_error("psg_ctx_create: device %d is %s; libpsg is built for gfx950 (MI355X) only", device,
                       prop.gcnArchName);
        return PSG_ERR_ARG;
    }
    PSG_CHECK_HIP(hipSetDevice(device));
    psg_ctx *c = new psg_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    *out = c;
    return PSG_OK;
}

extern "C" int psg_ctx_destroy(psg_ctx *ctx)
{
    delete ctx;
    return PSG_OK;
}
