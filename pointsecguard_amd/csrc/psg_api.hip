// Context, error reporting and version of libpsg.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "psg_common.h"

namespace {
thread_local char g_err[512] = "";
}

namespace psg {
bool trace_sync_enabled()
{
    static const bool on = getenv("PSG_TRACE_SYNC") && atoi(getenv("PSG_TRACE_SYNC"));
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
