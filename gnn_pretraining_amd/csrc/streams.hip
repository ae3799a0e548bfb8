// HIP streams vs hardware queues.  The ROCm runtime multiplexes a process's HIP streams onto a few hardware (AQL) queues
// (GPU_MAX_HW_QUEUES, 4 by default), binding a stream to a queue at the stream's first use, and packets of one queue run in
// order: two "concurrent" streams that landed on the same queue serialise.  Which queue a stream gets depends on what else
// the process created before, so the step executor does not guess: it MEASURES, once at start-up, which of its candidate
// streams really run beside each other (gmp_streams_share_queue) and builds its four-stream layout from streams that do.
#include "gnnmp_internal.h"

namespace {

// busy-wait on the constant-rate (100 MHz) wall clock; the iteration cap is the exit every wave reaches whatever the clock does
__global__ void spin_kernel(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    for (int i = 0; i < (1 << 22); ++i) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

}  // namespace

extern "C" int gmp_spin_us(int us, gmp_stream_t st) {
    if (us < 0 || us > 100000) return gmp::fail(GMP_ERR_ARG, "spin_us: %d us not in [0, 100000]", us);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, (uint64_t)us * 100);
    return gmp::check_launch("spin_kernel");
}

extern "C" int gmp_streams_share_queue(gmp_stream_t a_, gmp_stream_t b_, int* share) {
    if (!share) return gmp::fail(GMP_ERR_ARG, "streams_share_queue: null result pointer");
    hipStream_t a = (hipStream_t)a_, b = (hipStream_t)b_;
    if (a == b) {
        *share = 1;
        return GMP_OK;
    }
    hipEvent_t e0, ea, eb;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess)
        return gmp::fail(GMP_ERR_LAUNCH, "streams_share_queue: hipEventCreate failed");
    // first use binds either stream to its queue; drain both so the measurement starts from idle queues
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, (uint64_t)0);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, (uint64_t)0);
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    constexpr int SPIN_US = 400;
    (void)hipEventRecord(e0, a);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, (uint64_t)SPIN_US * 100);
    (void)hipEventRecord(ea, a);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, (uint64_t)0);      // behind the spin if a and b share a queue
    (void)hipEventRecord(eb, b);
    int rc = gmp::check_launch("streams_share_queue");
    float ta = 0.f, tb = 0.f;
    if (hipEventSynchronize(ea) != hipSuccess || hipEventSynchronize(eb) != hipSuccess || hipEventElapsedTime(&ta, e0, ea) != hipSuccess ||
        hipEventElapsedTime(&tb, e0, eb) != hipSuccess)
        rc = gmp::fail(GMP_ERR_LAUNCH, "streams_share_queue: event timing failed");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    if (rc == GMP_OK) *share = tb > 0.5f * ta ? 1 : 0;
    return rc;
}
