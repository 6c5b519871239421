// Shader-clock probe: one wavefront spins for ~0.3 ms and reports core cycles per 100 MHz reference tick.
// Run it in its own process beside a workload to see what the chip clocks to under that load:
//   hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe && ./clock_probe <seconds> [period_ms]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ void k_probe(unsigned long long *out, unsigned long long ticks) {
    const unsigned long long r0 = wall_clock64();
    const unsigned long long c0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = wall_clock64();
    }
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    const int period_ms = argc > 2 ? atoi(argv[2]) : 20;
    unsigned long long *d, h[2];
    if (hipMalloc(&d, 16) != hipSuccess) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (t > seconds) break;
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, 30000ull);  // 0.3 ms of the 100 MHz counter
        if (hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        printf("%8.3f s  %7.1f MHz\n", t, h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0);
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::milliseconds(period_ms));
    }
    hipFree(d);
    return 0;
}
