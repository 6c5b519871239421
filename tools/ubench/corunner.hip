// A synthetic neighbour for co-residency experiments: run it in its own process beside a workload and see what the
// workload's kernels lose.   ./corunner <mode> <seconds> [waves]
//   mode 0: one dependent f64 fma chain per wave, tiny loop (what a walk wavefront looks like to the issue logic)
//   mode 1: the same work spread over ~48 KB of straight-line code (instruction-cache footprint)
//   mode 2: a stream of 16-byte stores over 256 MB (memory traffic, next to no VALU)
//   hipcc --offload-arch=gfx950 -O2 corunner.hip -o corunner
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_chain(double *out, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    double x = 0.3 + threadIdx.x * 1e-9, s = 1.0000001;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll 1
        for (int i = 0; i < 64; i++)
            x = __fma_rn(x, s, 1e-9);
    }
    if (x == 12345.0)
        out[0] = x;
}

#define R8(a) a a a a a a a a
__global__ void k_bigcode(double *out, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    double x = 0.3 + threadIdx.x * 1e-9, s = 1.0000001;
    while (wall_clock64() - t0 < ticks) {
        // 8^4 = 4096 dependent fmas of 8 bytes each, plus the constant: ~48 KB of code walked front to back
        R8(R8(R8(R8(x = __fma_rn(x, s, 1e-9);))))
    }
    if (x == 12345.0)
        out[0] = x;
}

__global__ void k_stores(uint4 *buf, size_t n16, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    while (wall_clock64() - t0 < ticks) {
        buf[i % n16] = make_uint4(1, 2, 3, (unsigned)i);
        i += stride;
    }
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
    const int waves = argc > 3 ? atoi(argv[3]) : 512;
    double *d;
    uint4 *buf;
    const size_t n16 = (256u << 20) / 16;
    if (hipMalloc(&d, 64) != hipSuccess || hipMalloc(&buf, n16 * 16) != hipSuccess)
        return 1;
    const unsigned long long ticks = 2000000ull; /* 20 ms of the 100 MHz counter per launch */
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        if (mode == 0)
            hipLaunchKernelGGL(k_chain, dim3(waves), dim3(64), 0, 0, d, ticks);
        else if (mode == 1)
            hipLaunchKernelGGL(k_bigcode, dim3(waves), dim3(64), 0, 0, d, ticks);
        else
            hipLaunchKernelGGL(k_stores, dim3(waves), dim3(64), 0, 0, buf, n16, ticks);
        if (hipDeviceSynchronize() != hipSuccess)
            return 2;
        launches++;
    }
    printf("corunner mode %d: %d launches of %d waves\n", mode, launches, waves);
    return 0;
}
