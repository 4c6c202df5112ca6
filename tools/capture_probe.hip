// Which call of ANOTHER host thread invalidates a hipStreamCaptureModeThreadLocal capture?  (the lost hipGraph capture
// of the twelve-thread one-call-per-room NU leg, round 4.)  Thread A begins a capture on a stream that is either BLOCKING
// (hipStreamCreate) or NON-BLOCKING (hipStreamCreateWithFlags(hipStreamNonBlocking): what torch.cuda.Stream() hands out),
// launches a kernel, waits while thread B performs ONE operation, launches again and ends the capture.  Reported per
// (stream kind, operation): what B's call returned, what A's second launch / hipGetLastError / hipStreamEndCapture returned.
//   hipcc --offload-arch=gfx950 -O2 -o capture_probe tools/capture_probe.hip -lpthread && ./capture_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

__global__ void touch_kernel(float *p) { p[threadIdx.x] += 1.0f; }

struct Op { const char *name; std::function<hipError_t()> run; };

int main()
{
    float *d = nullptr, *d2 = nullptr;
    hipMalloc(&d, 4096);
    hipMalloc(&d2, 4096);
    hipMemset(d, 0, 4096);
    static float host[1024];
    float *pinned = nullptr;
    hipHostMalloc(&pinned, 4096, 0);
    hipStream_t other_nb = nullptr, other_bl = nullptr;
    hipStreamCreateWithFlags(&other_nb, hipStreamNonBlocking);
    hipStreamCreate(&other_bl);
    hipEvent_t ev_other = nullptr;
    hipEventCreateWithFlags(&ev_other, hipEventDisableTiming);
    // an event recorded (eagerly) on a stream of B before A starts capturing, polled during the capture: what torch's
    // pinned-host allocator does
    hipEventRecord(ev_other, other_nb);
    hipDeviceSynchronize();

    std::vector<Op> ops = {
        {"nothing", [&] { return hipSuccess; }},
        {"hipMemcpy H2D pageable (legacy stream)", [&] { return hipMemcpy(d2, host, 4096, hipMemcpyHostToDevice); }},
        {"hipMemcpy D2H pageable (legacy stream)", [&] { return hipMemcpy(host, d2, 4096, hipMemcpyDeviceToHost); }},
        {"hipMemcpyAsync H2D pinned on stream 0", [&] { return hipMemcpyAsync(d2, pinned, 4096, hipMemcpyHostToDevice, 0); }},
        {"hipMemset (legacy stream)", [&] { return hipMemset(d2, 0, 4096); }},
        {"kernel launch on stream 0", [&] { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, 0, d2); return hipGetLastError(); }},
        {"hipMemcpyAsync H2D pageable on own non-blocking stream + sync",
         [&] { hipError_t e = hipMemcpyAsync(d2, host, 4096, hipMemcpyHostToDevice, other_nb); return e != hipSuccess ? e : hipStreamSynchronize(other_nb); }},
        {"kernel on own BLOCKING stream + sync",
         [&] { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, other_bl, d2); hipError_t e = hipGetLastError(); return e != hipSuccess ? e : hipStreamSynchronize(other_bl); }},
        {"hipMalloc + hipFree", [&] { void *p = nullptr; hipError_t e = hipMalloc(&p, 1 << 20); return e != hipSuccess ? e : hipFree(p); }},
        {"hipHostMalloc + hipHostFree", [&] { void *p = nullptr; hipError_t e = hipHostMalloc(&p, 1 << 20, 0); return e != hipSuccess ? e : hipHostFree(p); }},
        {"hipStreamCreate(non-blocking) + destroy", [&] { hipStream_t s; hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking); return e != hipSuccess ? e : hipStreamDestroy(s); }},
        {"hipFuncSetAttribute", [&] { return hipFuncSetAttribute((const void *)touch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); }},
        {"hipEventQuery of an event of B's own stream", [&] { hipError_t e = hipEventQuery(ev_other); return e == hipErrorNotReady ? hipSuccess : e; }},
        {"hipDeviceSynchronize", [&] { return hipDeviceSynchronize(); }},
        {"hipStreamSynchronize(0)", [&] { return hipStreamSynchronize(0); }},
        {"hipStreamBeginCapture(ThreadLocal)+kernel+EndCapture on own non-blocking stream",
         [&] {
             hipError_t e = hipStreamBeginCapture(other_nb, hipStreamCaptureModeThreadLocal);
             if (e != hipSuccess) return e;
             hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, other_nb, d2);
             hipGraph_t g = nullptr;
             e = hipStreamEndCapture(other_nb, &g);
             if (g) hipGraphDestroy(g);
             return e;
         }},
    };

    for (int blocking = 0; blocking < 2; ++blocking) {
        printf("== capturing stream: %s ==\n", blocking ? "BLOCKING (hipStreamCreate)" : "NON-BLOCKING (torch.cuda.Stream)");
        for (const Op &op : ops) {
            hipStream_t st = nullptr;
            if (blocking) hipStreamCreate(&st); else hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            (void)hipGetLastError();
            hipError_t e_begin = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, st, d);
            hipError_t e_l1 = hipGetLastError();
            hipError_t e_b = hipSuccess;
            std::thread tb([&] { hipSetDevice(0); e_b = op.run(); (void)hipGetLastError(); });
            tb.join();
            hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, st, d);
            hipError_t e_l2 = hipGetLastError();
            hipGraph_t g = nullptr;
            hipError_t e_end = hipStreamEndCapture(st, &g);
            printf("%-78s B: %-28s | A: begin %s, launch1 %s, launch2 %s, end %s%s\n", op.name, hipGetErrorName(e_b), hipGetErrorName(e_begin),
                   hipGetErrorName(e_l1), hipGetErrorName(e_l2), hipGetErrorName(e_end), g ? "" : " (no graph)");
            if (g) hipGraphDestroy(g);
            (void)hipGetLastError();
            hipStreamDestroy(st);
            hipDeviceSynchronize();
            (void)hipGetLastError();
        }
    }
    // the same operations from the CAPTURING thread itself (what PSG_LAUNCH_CHECK's hipGetLastError and a stray
    // hipFuncSetAttribute / hipMemcpyAsync do inside a capture)
    printf("== same thread, non-blocking capturing stream ==\n");
    for (const Op &op : ops) {
        hipStream_t st = nullptr;
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        (void)hipGetLastError();
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, st, d);
        hipError_t e_l1 = hipGetLastError();
        hipError_t e_b = op.run();
        (void)hipGetLastError();
        hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, st, d);
        hipError_t e_l2 = hipGetLastError();
        hipGraph_t g = nullptr;
        hipError_t e_end = hipStreamEndCapture(st, &g);
        printf("%-78s op: %-27s | launch1 %s, launch2 %s, end %s%s\n", op.name, hipGetErrorName(e_b), hipGetErrorName(e_l1), hipGetErrorName(e_l2),
               hipGetErrorName(e_end), g ? "" : " (no graph)");
        if (g) hipGraphDestroy(g);
        (void)hipGetLastError();
        hipStreamDestroy(st);
        hipDeviceSynchronize();
        (void)hipGetLastError();
    }
    return 0;
}
