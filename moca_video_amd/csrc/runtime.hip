// Runtime helpers of libmoca_hip.so: hipGraph capture/replay of a launch sequence,
// streams, events and device query.  A UNet forward is ~1000 short launches; after the
// first eager pass the host replays it as one hipGraph (no tracing compiler involved:
// the graph is just the recorded launch sequence of the C-ABI calls above).
#include "common.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

extern "C" int moca_graph_begin(void* stream) {
    if (hipStreamBeginCapture(moca_stream(stream), hipStreamCaptureModeThreadLocal) != hipSuccess) return MOCA_E_GRAPH;
    return MOCA_OK;
}

extern "C" int moca_graph_end(void* stream, void** graph_exec_out) {
    if (!graph_exec_out) return MOCA_E_BADARG;
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(moca_stream(stream), &g) != hipSuccess || !g) return MOCA_E_GRAPH;
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess || !ge) return MOCA_E_GRAPH;
    *graph_exec_out = reinterpret_cast<void*>(ge);
    return MOCA_OK;
}

extern "C" int moca_graph_launch(void* graph_exec, void* stream) {
    if (!graph_exec) return MOCA_E_BADARG;
    if (hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), moca_stream(stream)) != hipSuccess) return MOCA_E_GRAPH;
    return MOCA_OK;
}

extern "C" int moca_graph_destroy(void* graph_exec) {
    if (!graph_exec) return MOCA_OK;
    return hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)) == hipSuccess ? MOCA_OK : MOCA_E_GRAPH;
}

extern "C" int moca_stream_create(void** stream_out) {
    if (!stream_out) return MOCA_E_BADARG;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return MOCA_E_LAUNCH;
    *stream_out = reinterpret_cast<void*>(s);
    return MOCA_OK;
}
extern "C" int moca_stream_destroy(void* stream) {
    return hipStreamDestroy(moca_stream(stream)) == hipSuccess ? MOCA_OK : MOCA_E_LAUNCH;
}
extern "C" int moca_stream_sync(void* stream) {
    return hipStreamSynchronize(moca_stream(stream)) == hipSuccess ? MOCA_OK : MOCA_E_LAUNCH;
}

extern "C" int moca_event_create(void** ev_out) {
    if (!ev_out) return MOCA_E_BADARG;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return MOCA_E_LAUNCH;
    *ev_out = reinterpret_cast<void*>(e);
    return MOCA_OK;
}
extern "C" int moca_event_record(void* ev, void* stream) {
    return hipEventRecord(reinterpret_cast<hipEvent_t>(ev), moca_stream(stream)) == hipSuccess ? MOCA_OK : MOCA_E_LAUNCH;
}
extern "C" int moca_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
    if (!ms_out) return MOCA_E_BADARG;
    if (hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev_stop)) != hipSuccess) return MOCA_E_LAUNCH;
    return hipEventElapsedTime(ms_out, reinterpret_cast<hipEvent_t>(ev_start), reinterpret_cast<hipEvent_t>(ev_stop)) == hipSuccess
               ? MOCA_OK : MOCA_E_LAUNCH;
}
extern "C" int moca_event_destroy(void* ev) {
    return hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)) == hipSuccess ? MOCA_OK : MOCA_E_LAUNCH;
}

extern "C" int moca_device_info(char* name, int32_t len, int32_t* cu_count) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MOCA_E_NODEVICE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MOCA_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return MOCA_E_NODEVICE;
    if (name && len > 0) {
        strncpy(name, prop.gcnArchName, (size_t)len - 1);
        name[len - 1] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    return MOCA_OK;
}

// (a kernel, not hipMemsetAsync: with the library loaded before torch the process holds two HIP runtime images and the memset
//  entry of ours reports "no ROCm-capable device" while kernel launches resolve fine -- seen with build() + smoke() in one process)
__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p, int64_t n16, unsigned char* __restrict__ tail, int ntail) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = uint4{0u, 0u, 0u, 0u};
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

extern "C" int moca_memset_zero(void* ptr, int64_t bytes, void* stream) {
    if (!ptr || bytes < 0 || (reinterpret_cast<uintptr_t>(ptr) & 15)) return MOCA_E_BADARG;
    if (bytes == 0) return MOCA_OK;
    const int64_t n16 = bytes / 16;
    int blocks = (int)((n16 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, moca_stream(stream), reinterpret_cast<uint4*>(ptr), n16,
                       reinterpret_cast<unsigned char*>(ptr) + n16 * 16, (int)(bytes - n16 * 16));
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

// ---- measurement aid (tools/clock_in_step.py): what shader clock does the chip hold WHILE another stream's launches run?  `blocks` single-
// wave blocks (one per XCD with blocks = 8: consecutive workgroups go to consecutive XCDs) each record `nsamples` pairs
// (s_memtime = shader cycles, s_memrealtime = the constant 100 MHz counter) about 8 us apart; between two samples the clock is
// d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back (6)).  A block occupies one wave slot, no LDS, 8 registers.
// Bounded by nsamples (and an optional stop flag); results: u64 [blocks][nsamples][2], zeroed by the caller.
__global__ __launch_bounds__(64) void clock_sampler_kernel(unsigned long long* __restrict__ buf, int nsamples, const volatile int* stop) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x) return;
    unsigned long long* b = buf + (size_t)blockIdx.x * nsamples * 2;
    for (int i = 0; i < nsamples; ++i) {
        unsigned long long t, r;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r)::"memory");
        b[2 * i] = t;
        b[2 * i + 1] = r;
        if (stop && *stop) break;
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
    }
#endif
}
extern "C" int moca_debug_clock_sampler(void* buf, int32_t blocks, int32_t nsamples, const int32_t* stop, void* stream) {
    if (!buf || blocks < 1 || blocks > 64 || nsamples < 1 || nsamples > (1 << 20)) return MOCA_E_BADARG;
    hipLaunchKernelGGL(clock_sampler_kernel, dim3(blocks), dim3(64), 0, moca_stream(stream), reinterpret_cast<unsigned long long*>(buf), nsamples,
                       reinterpret_cast<const volatile int*>(stop));
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

// ---- kernel-choice knobs for tests / A-B runs (never results): one table instead of getenv() calls in the launchers
static int g_tuning[MOCA_TUNE_COUNT] = {1, 1, 1, 1, 1, 1, 0, 1, 0, 0, 1};
int moca_tuning_get(int knob) { return (knob >= 0 && knob < MOCA_TUNE_COUNT) ? g_tuning[knob] : 0; }
extern "C" int moca_set_tuning(int32_t knob, int32_t value) {
    if (knob < 0 || knob >= MOCA_TUNE_COUNT || value < 0 || value > 2) return MOCA_E_BADARG;
    const int old = g_tuning[knob];
    g_tuning[knob] = value;
    return old;
}

