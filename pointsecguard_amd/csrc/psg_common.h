// Internal helpers shared by the libpsg translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/psg.h"

namespace psg {

void set_error(const char *fmt, ...);

// Blocking host <-> device copy that does NOT go through the legacy default stream: hipMemcpy synchronises with every
// blocking stream of the process, which is an error - for the copying thread AND for the capture - while another host
// thread captures a hipGraph on one ("operation would make the legacy stream depend on a capturing blocking stream":
// the one-call-per-room NU harness builds its model copies from twelve threads).  A non-blocking stream of its own per call.
hipError_t copy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
// The same for hipMemset.  Measured (tools/capture_probe.hip, round 5): ANOTHER host thread's hipMemcpy / hipMemset on the
// legacy stream - and its hipDeviceSynchronize - INVALIDATE a hipStreamCaptureModeThreadLocal capture in progress on any
// stream of the process, blocking or not (the caller gets hipErrorStreamCaptureImplicit / ...Unsupported, the capturing
// thread's next launch hipErrorStreamCaptureInvalidated); hipMalloc / hipFree, stream and event creation,
// hipFuncSetAttribute, async work on stream 0 and hipStreamSynchronize do not.  So no entry point that can run beside a
// capture (creation, upload, forward / backward / attack) uses the legacy-stream forms; only the test read-backs
// (psg_*_debug_read, psg_gcn_knn_stats, psg_rla_sampler_possibility) synchronise the device.
hipError_t memset_sync(void *dst, int value, size_t bytes);

#define PSG_CHECK_HIP(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            psg::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PSG_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define PSG_REQUIRE(cond, ...)                                                                   \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            psg::set_error(__VA_ARGS__);                                                         \
            return PSG_ERR_ARG;                                                                  \
        }                                                                                        \
    } while (0)

// Every environment switch the library reads goes through these two (psg_api.hip): they return the variable's value and
// record its NAME, so that psg_env_switches() can tell a caller (bench.py prints them and refuses the result-changing
// ones) which non-default switches this process runs under.  Switches select between tested paths or turn on diagnosis;
// the only ones that change RESULTS exist in -DPSG_DIAG_BUILD libraries.
const char *env_str(const char *name);
int env_int(const char *name, int dflt);

// Diagnosis only (PSG_TRACE_SYNC=1, eager launches outside any stream capture): after every checked launch the device is
// synchronised and the launch site is written to stderr, so that a run that stops shows the last launch that completed
// and the one that did not (tools/profile_round.sh gmfma_trace; DESIGN.md section 4, "the counter-pass hang").
bool trace_sync_enabled();
void trace_sync_point(const char *file, int line);

#define PSG_LAUNCH_CHECK()                                                                       \
    do {                                                                                         \
        PSG_CHECK_HIP(hipGetLastError());                                                        \
        if (psg::trace_sync_enabled()) psg::trace_sync_point(__FILE__, __LINE__);                 \
    } while (0)

// Every model / workspace handle gets a process-unique, never re-used generation number at creation.  The hipGraph keys of the
// replayed loops compare THESE, not handle addresses: the allocator may hand a new handle the address of a freed one, and a
// graph keyed on the address would then replay kernels whose arguments point into the freed arena (advisor, round 4).
uint64_t next_generation();
uint64_t pn2_model_generation(const psg_pn2_model *m);
int pn2_forward_lean(psg_pn2_model *m, psg_pn2_ws *ws, int fwd, const float *x0, float *logp_out, psg_stream stream);
// psg_pn2_backward for the colour channels of the input gradient only (the NU loop; psg_pn2.hip)
int pn2_backward_colour(psg_pn2_model *m, psg_pn2_ws *ws, int fwd, const float *dlogp, float *dx0_out, psg_stream stream);
uint64_t pn2_ws_generation(const psg_pn2_ws *ws);

// hipGraph bookkeeping of the three replayed loops (psg_pn2_nu_window, psg_gcn_nb_attack, psg_rla_bim_attack): a capture that
// fails falls back to the eager launches - correct, but slower - so it is COUNTED, per handle and for the process
// (psg_capture_stats), and a handle whose capture failed does not try again for the same key.
struct CaptureCounters { long long tried = 0, failed = 0, replays = 0, eager = 0; };
void capture_note(CaptureCounters *own, int tried, int failed, int replays, int eager);

// Optional per-launch HIP-event timing of a workspace (psg_*_prof_enable / psg_*_prof_read): pairs of events recorded on
// the LAUNCH stream around a launch (or a group of launches) with a tag and the algorithmic FLOPs of that launch; off in
// normal operation (one branch per launch).  bench.py's `roofline` objects are computed from these.
struct EvLog {
    bool on = false;
    std::vector<hipEvent_t> ev;      // pairs
    std::vector<int> tag;
    std::vector<double> flop, bytes;     // algorithmic FLOPs / bytes of the tagged launch
    size_t used = 0;
    void reset(bool enable) { on = enable; used = 0; tag.clear(); flop.clear(); bytes.clear(); }
    void destroy() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); ev.clear(); }
    // sums per tag; blocks until the events have completed
    int read(int n_tags, double *total_ms, int *counts, double *flops, double *bytes_out = nullptr)
    {
        for (int i = 0; i < n_tags; ++i) { total_ms[i] = 0.0; counts[i] = 0; if (flops) flops[i] = 0.0; if (bytes_out) bytes_out[i] = 0.0; }
        for (size_t i = 0; i < tag.size(); ++i) {
            float ms = 0.f;
            if (hipEventSynchronize(ev[2 * i + 1]) != hipSuccess) return -1;
            if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) return -1;
            if (tag[i] < n_tags) {
                total_ms[tag[i]] += ms; counts[tag[i]] += 1;
                if (flops) flops[tag[i]] += flop[i];
                if (bytes_out) bytes_out[tag[i]] += bytes[i];
            }
        }
        return 0;
    }
};
struct EvScope {
    EvLog *log; hipStream_t st; hipEvent_t stop = nullptr;
    EvScope(EvLog *l, int tag, double flop, hipStream_t s, double bytes = 0.0) : log(l), st(s)
    {
        if (!log || !log->on) { log = nullptr; return; }
        while (log->used + 2 > log->ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { log = nullptr; return; }
            log->ev.push_back(e);
        }
        hipEvent_t start = log->ev[log->used];
        stop = log->ev[log->used + 1];
        log->used += 2;
        log->tag.push_back(tag);
        log->flop.push_back(flop);
        log->bytes.push_back(bytes);
        (void)hipEventRecord(start, st);
    }
    ~EvScope() { if (log) (void)hipEventRecord(stop, st); }
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace psg

struct psg_ctx {
    int device;
    int num_cu;
};
