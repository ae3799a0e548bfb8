// Probe (round 3): what does replaying a step-shaped hipGraph cost on this runtime, host side and GPU side, against enqueueing the same
// kernels eagerly?  A step is ~250 short kernels on 4 streams with ~40 cross-stream dependencies.  Build: hipcc --offload-arch=gfx950 -O2
// scripts/probe_graph.hip -o gpurun_out/probe_graph; run on the GPU box.  Prints one line per variant.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(float* p, int spin_ns) {          // ~spin_ns of wall time per block, 128 blocks
    const long long t0 = wall_clock64();               // 100 MHz ticks
    while ((wall_clock64() - t0) * 10 < spin_ns) {}
    if (threadIdx.x == 0 && p) p[blockIdx.x] += 1.f;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the shape: main chain of `layers` x 5 kernels; aux gets 2 kernels per layer depending on main's 2nd and 4th kernel of that layer;
// two head streams fork after the first third, run 25 kernels each, join before the second half
static int enqueue(hipStream_t* s, hipEvent_t* ev, float* buf, int layers, int spin, int* nk, int* ndep) {
    int e = 0, k = 0;
    auto L = [&](hipStream_t st) { hipLaunchKernelGGL(work, dim3(128), dim3(64), 0, st, buf, spin); ++k; };
    for (int l = 0; l < layers; ++l) {
        for (int i = 0; i < 5; ++i) {
            L(s[0]);
            if (i == 1 || i == 3) {
                (void)hipEventRecord(ev[e], s[0]); (void)hipStreamWaitEvent(s[1], ev[e], 0); ++e;
                L(s[1]);
            }
        }
        if (l == layers / 3) {
            (void)hipEventRecord(ev[e], s[0]);
            for (int h = 2; h < 4; ++h) { (void)hipStreamWaitEvent(s[h], ev[e], 0); for (int i = 0; i < 25; ++i) L(s[h]); }
            ++e;
            for (int h = 2; h < 4; ++h) { (void)hipEventRecord(ev[e], s[h]); (void)hipStreamWaitEvent(s[0], ev[e], 0); ++e; }
        }
    }
    (void)hipEventRecord(ev[e], s[1]); (void)hipStreamWaitEvent(s[0], ev[e], 0); ++e;
    *nk = k; *ndep = e;
    return 0;
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 30, spin = argc > 2 ? atoi(argv[2]) : 4000, reps = 200;
    hipStream_t s[4];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(512);
    for (auto& x : ev) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    float* buf;
    CK(hipMalloc(&buf, 4096));
    CK(hipMemset(buf, 0, 4096));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    int nk = 0, nd = 0;
    // ---- eager
    for (int w = 0; w < 3; ++w) enqueue(s, ev.data(), buf, layers, spin, &nk, &nd);
    CK(hipDeviceSynchronize());
    double host = 0;
    CK(hipEventRecord(t0, s[0]));
    for (int r = 0; r < reps; ++r) { const double a = now_us(); enqueue(s, ev.data(), buf, layers, spin, &nk, &nd); host += now_us() - a; }
    CK(hipEventRecord(t1, s[0]));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, t0, t1));
    printf("eager : %d kernels, %d cross-stream deps, spin %d ns: host %.1f us/step (%.2f us/kernel), gpu %.1f us/step, ideal chain %.1f us\n", nk, nd, spin,
           host / reps, host / reps / nk, ms * 1e3 / reps, layers * 5 * spin * 1e-3);
    // ---- graph: capture the same enqueue
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal));
    enqueue(s, ev.data(), buf, layers, spin, &nk, &nd);
    CK(hipStreamEndCapture(s[0], &g));
    size_t nnodes = 0;
    CK(hipGraphGetNodes(g, nullptr, &nnodes));
    double a = now_us();
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const double inst = now_us() - a;
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s[0]));
    CK(hipDeviceSynchronize());
    host = 0;
    CK(hipEventRecord(t0, s[0]));
    for (int r = 0; r < reps; ++r) { a = now_us(); CK(hipGraphLaunch(ge, s[0])); host += now_us() - a; }
    CK(hipEventRecord(t1, s[0]));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, t0, t1));
    printf("graph : %zu nodes, instantiate %.0f us: host %.1f us/replay, gpu %.1f us/replay\n", nnodes, inst, host / reps, ms * 1e3 / reps);
    // ---- graph, single chain (all on one stream): the in-order floor
    hipGraph_t g1; hipGraphExec_t ge1;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal));
    for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(work, dim3(128), dim3(64), 0, s[0], buf, spin);
    CK(hipStreamEndCapture(s[0], &g1));
    CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge1, s[0]));
    CK(hipDeviceSynchronize());
    host = 0;
    CK(hipEventRecord(t0, s[0]));
    for (int r = 0; r < reps; ++r) { a = now_us(); CK(hipGraphLaunch(ge1, s[0])); host += now_us() - a; }
    CK(hipEventRecord(t1, s[0]));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, t0, t1));
    printf("graph1: %d kernels in one chain: host %.1f us/replay, gpu %.1f us/replay (%.2f us/kernel)\n", nk, host / reps, ms * 1e3 / reps, ms * 1e3 / reps / nk);
    // ---- eager single chain
    CK(hipEventRecord(t0, s[0]));
    host = 0;
    for (int r = 0; r < reps; ++r) { a = now_us(); for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(work, dim3(128), dim3(64), 0, s[0], buf, spin); host += now_us() - a; }
    CK(hipEventRecord(t1, s[0]));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, t0, t1));
    printf("eager1: %d kernels in one chain: host %.1f us/step, gpu %.1f us/step (%.2f us/kernel)\n", nk, host / reps, ms * 1e3 / reps, ms * 1e3 / reps / nk);
    return 0;
}
