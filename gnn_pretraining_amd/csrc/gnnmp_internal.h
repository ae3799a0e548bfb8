// Internal helpers shared by the libgnnmp kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gnnmp.h"

#define GMP_WAVE 64
#define GMP_MAX_GROUPS 24

namespace gmp {

// thread-local last-error text, read through gmp_last_error_string()
char* err_buf();
int fail(int code, const char* fmt, ...);

// The next gmp_gemm_f32 / gmp_gemm_f32_grouped launch made by this host thread stores `value` to *flag when its first workgroup
// starts (a cross-stream gate opened by the GEMM that follows the producer anyway, instead of by a launch of its own).  A GEMM
// call that returns without launching leaves the signal pending: the caller checks signal_pending() and opens the gate itself.
void signal_on_next_gemm(int32_t* flag, int value);
bool signal_pending();

// Two-lane enqueue (step.hip): the pre-training step's launch sequence is walked by TWO host threads at once -- the caller takes the main
// stream's launches, a worker thread everything else -- because one thread needs ~4 us per launch and a step has ~250 (the launcher was the
// limiter on slower hosts; two threads launching on different streams scale 1.7x on this runtime, scripts/probe_two_threads.hip).  Both
// threads run the SAME code; a launch (kernel, async memset / copy) on a stream the thread does not take is skipped, here, below every
// gmp_* entry point.  Thread-local; 0 = take everything (every caller outside the step executor).
enum { LANE_ALL = 0, LANE_ONLY = 1, LANE_ALL_BUT = 2 };
int& lane_mode();
hipStream_t& lane_stream();
inline bool lane_takes(hipStream_t st) {
    const int m = lane_mode();
    return m == LANE_ALL || ((m == LANE_ONLY) == (st == lane_stream()));
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GMP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GMP_OK;
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the DEVICE's copy of a function: a launcher keeps one "done" bit per device
// ordinal (std::atomic<uint64_t>, function-local static) instead of a process-wide bool -- a process driving a second GPU would otherwise
// skip the call there and fail to launch its > 64 KiB tiles.  Racing threads at worst both make the (idempotent) call; ordinals >= 64 repeat it.
inline int cur_device() { int d = 0; (void)hipGetDevice(&d); return d; }
inline bool lds_attr_done(const std::atomic<uint64_t>& m) { const int d = cur_device(); return d < 64 && ((m.load(std::memory_order_acquire) >> d) & 1ull); }
inline void lds_attr_mark(std::atomic<uint64_t>& m) { const int d = cur_device(); if (d < 64) m.fetch_or(1ull << d, std::memory_order_release); }

// ---- Philox4x32-10 (counter-based RNG for dropout; mask is regenerated in the
// backward pass from (seed, stream, element) instead of being stored) ----------
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32): written as __umulhi + a 32-bit product hipcc issued TWO quarter-rate multiplies
        // each -- 308 of them per thread in the BatchNorm forward that draws a mask for eight float4 (round 3)
        const uint64_t p0 = (uint64_t)M0 * ctr.x, p1 = (uint64_t)M1 * ctr.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}

// keep-mask for 4 consecutive elements starting at element index 4*q of dropout
// stream `stream`; returns the multiplier (0 or 1/(1-p)) per element.
__device__ __forceinline__ float4 dropout_scale4(uint64_t seed, uint32_t stream, uint64_t q, float p, float inv_keep) {
    uint4 r = philox4x32(make_uint4((uint32_t)q, (uint32_t)(q >> 32), stream, 0x6d70u),
                         make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    // u = r * 2^-32 in [0,1); keep iff u >= p
    const float s = 2.3283064365386963e-10f;
    float4 o;
    o.x = ((float)r.x * s >= p) ? inv_keep : 0.f;
    o.y = ((float)r.y * s >= p) ? inv_keep : 0.f;
    o.z = ((float)r.z * s >= p) ? inv_keep : 0.f;
    o.w = ((float)r.w * s >= p) ? inv_keep : 0.f;
    return o;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace gmp


// every launch of the library goes through the lane filter (see gmp::lane_takes)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                        \
    do {                                                                                                                         \
        if (gmp::lane_takes((hipStream_t)(streamId))) hipLaunchKernelGGLInternal((kernelName), (numBlocks), (numThreads), (memPerBlock), (streamId), __VA_ARGS__); \
    } while (0)
#define hipMemsetAsync(dst, value, bytes, stream) (gmp::lane_takes((hipStream_t)(stream)) ? hipMemsetAsync((dst), (value), (bytes), (stream)) : hipSuccess)
#define hipMemcpyAsync(dst, src, bytes, kind, stream) (gmp::lane_takes((hipStream_t)(stream)) ? hipMemcpyAsync((dst), (src), (bytes), (kind), (stream)) : hipSuccess)
#define hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, stream) \
    (gmp::lane_takes((hipStream_t)(stream)) ? hipMemcpy2DAsync((dst), (dpitch), (src), (spitch), (width), (height), (kind), (stream)) : hipSuccess)
