// Probe: does enqueueing from TWO host threads (each on its own stream) scale on this runtime?  hipcc --offload-arch=gfx950 -O2 -pthread
// scripts/probe_two_threads.hip -o gpurun_out/probe_two_threads; prints host microseconds per launch for one thread and for two at once.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <atomic>

__global__ void tiny(float* p, int spin_ns) {
    const long long t0 = wall_clock64();
    while ((wall_clock64() - t0) * 10 < spin_ns) {}
    if (threadIdx.x == 0 && p) p[blockIdx.x] += 1.f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s[4];
    for (auto& x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    float* buf; (void)hipMalloc(&buf, 4096); (void)hipMemset(buf, 0, 4096);
    const int N = 2000;
    for (int spin : {0, 3000}) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s[0], buf, spin);
        (void)hipDeviceSynchronize();
        double a = now_us();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s[0], buf, spin);
        double one = (now_us() - a) / N;
        (void)hipDeviceSynchronize();
        a = now_us();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s[i & 1], buf, spin);
        double alt = (now_us() - a) / N;
        (void)hipDeviceSynchronize();
        std::atomic<int> go{0};
        double t_thr[2];
        auto work = [&](int id) {
            while (!go.load()) {}
            double b = now_us();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s[id], buf, spin);
            t_thr[id] = (now_us() - b) / N;
        };
        std::thread th0(work, 0), th1(work, 1);
        double b = now_us();
        go.store(1);
        th0.join(); th1.join();
        double wall = now_us() - b;
        (void)hipDeviceSynchronize();
        printf("kernel ~%d ns: one thread %.2f us/launch; one thread over two streams %.2f; two threads %.2f / %.2f us per launch each, %d launches in %.0f us wall = %.2f us per launch overall\n",
               spin, one, alt, t_thr[0], t_thr[1], 2 * N, wall, wall / (2 * N));
    }
    return 0;
}
