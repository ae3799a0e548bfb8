// HIP streams vs hardware queues.  The ROCm runtime multiplexes a process's HIP streams onto a few hardware (AQL) queues
// (GPU_MAX_HW_QUEUES, 4 by default), binding a stream to a queue at the stream's first use, and packets of one queue run in
// order: two "concurrent" streams that landed on the same queue serialise.  Which queue a stream gets depends on what else
// the process created before, so the step executor does not guess: it MEASURES, once at start-up, which of its candidate
// streams really run beside each other (gmp_streams_share_queue) and builds its four-stream layout from streams that do.
#include <stdlib.h>

#include <atomic>

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

// Device-side gates: the cross-stream dependencies of the step executor without barrier packets.  Measured on MI355X
// (scripts/diag_blocked_queues.py): every hardware queue parked on a hipStreamWaitEvent adds ~2 us to EVERY kernel boundary of
// the queues that are running (0.8 us -> 2.7 / 4.2 / 7.0 us with one / two / three parked queues), and in a step the host runs
// ahead, so two or three of the four queues are parked most of the time.  A gate is one wave that sleeps on flag words until
// all of them have reached `want`; the stream behind it is held by an ordinary running kernel and nobody else pays
// (10.83 vs 10.80 us per kernel with three gated streams).  ONLY for streams on different hardware queues: a gate in front of
// its own opener in one in-order queue would wait for its time-out.  A gate that times out (two minutes) sets *err.
constexpr int GATE_FLAGS = 64;
__global__ void gate_wait_kernel(const int* flags, unsigned long long mask, int want, int* err, unsigned long long timeout_ticks) {
    const int lane = threadIdx.x;
    const bool mine = (mask >> lane) & 1ull;
    bool ok = !mine;
    const unsigned long long t0 = wall_clock64();
    // exit every wave reaches: all flags seen, or the wall clock (checked every 1,024 polls) says the opener is not coming --
    // two minutes by default (GMP_GATE_TIMEOUT_S; the tests use a few seconds), because in a data-parallel run the stream that
    // opens this gate may itself be waiting for a peer GPU
    for (unsigned i = 1;; ++i) {
        if (!ok) ok = __hip_atomic_load(flags + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want;
        if (__all(ok)) return;
        __builtin_amdgcn_s_sleep(8);
        if ((i & 1023u) == 0 && wall_clock64() - t0 > timeout_ticks) break;
    }
    if (!ok && err) atomicOr(err, 1);
}
__global__ void gate_open_kernel(int* flag, int value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace

// time-out of a gate in ticks of the 100 MHz wall clock: GMP_GATE_TIMEOUT_S (fractions allowed) or two minutes, until
// gmp_gate_set_timeout replaces it (the data-parallel start-up check of the engine shortens it for its probe steps)
static unsigned long long ticks_of(double seconds) {
    if (!(seconds > 0.0)) seconds = 120.0;
    if (seconds > 3600.0) seconds = 3600.0;
    return (unsigned long long)(seconds * 1e8);
}
static std::atomic<unsigned long long>& gate_ticks() {
    static std::atomic<unsigned long long> t{ticks_of(getenv("GMP_GATE_TIMEOUT_S") ? atof(getenv("GMP_GATE_TIMEOUT_S")) : 120.0)};
    return t;
}
extern "C" int gmp_gate_set_timeout(double seconds) {
    if (!(seconds > 0.0) || seconds > 3600.0) return gmp::fail(GMP_ERR_ARG, "gate_set_timeout: %g s not in (0, 3600]", seconds);
    gate_ticks().store(ticks_of(seconds));
    return GMP_OK;
}

extern "C" int gmp_gate_wait(const int32_t* flags, uint64_t mask, int want, int32_t* err, gmp_stream_t st) {
    if (!flags) return gmp::fail(GMP_ERR_ARG, "gate_wait: null flags");
    if (!mask) return GMP_OK;
    const unsigned long long ticks = gate_ticks().load();
    hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(GATE_FLAGS), 0, (hipStream_t)st, (const int*)flags, (unsigned long long)mask, want, (int*)err, ticks);
    return gmp::check_launch("gate_wait_kernel");
}
extern "C" int gmp_gate_open(int32_t* flag, int value, gmp_stream_t st) {
    if (!flag) return gmp::fail(GMP_ERR_ARG, "gate_open: null flag");
    hipLaunchKernelGGL(gate_open_kernel, dim3(1), dim3(1), 0, (hipStream_t)st, (int*)flag, value);
    return gmp::check_launch("gate_open_kernel");
}

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

// *word += inc, by one thread, in stream order: the per-replay dropout seed of a captured step (gmp_bn_config.seed_dev)
namespace {
__global__ void counter_add_kernel(uint64_t* word, uint64_t inc) { *word += inc; }
}  // namespace
extern "C" int gmp_counter_add(uint64_t* word, uint64_t inc, gmp_stream_t stream) {
    if (!word) return gmp::fail(GMP_ERR_ARG, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, word, inc);
    return gmp::check_launch("counter_add_kernel");
}
