// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the
// synthesis walk is made of, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2048
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double s, int n)
{
    double a[8];
    int b[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001 + i; b[i] = threadIdx.x + i; }
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __dadd_rn(a[i], s);
            if (OP == 1) { b[i] += (int)a[i]; a[i] = __hiloint2double(__double2hiint(a[i]) ^ 1, __double2loint(a[i])); }
            if (OP == 2) b[i] = (b[i] << 2) + b[(i + 1) & 7];
            if (OP == 3) b[i] = b[i] ^ b[(i + 3) & 7];
            if (OP == 4) { asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 5) a[i] = __dmul_rn(a[i], s);
            if (OP == 6) b[i] = __double2hiint(a[i]) >= 0x408FF800 ? b[i] + 1 : b[i];
            if (OP == 7) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b[i]) : "v"(a[i])); }
        }
    }
    double r = 0;
    for (int i = 0; i < 8; i++) r += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char *name, int instr_per_iter)
{
    double *d;
    const int blocks = 256 * 8, threads = 256; // 16 waves/CU = 4 per SIMD
    hipMalloc(&d, blocks * threads * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 1e-9, 16);
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, 1e-9, ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD processed sequentially: total waves = blocks*4; per SIMD = blocks*4/1024
    const double wave_instr_per_simd = (double)blocks * 4 / 1024 * ITER * 8 * instr_per_iter;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz nominal)\n", name, ms, cycles / wave_instr_per_simd);
    hipFree(d);
}

int main()
{
    run<0>("v_add_f64", 1);
    run<5>("v_mul_f64", 1);
    run<7>("v_cvt_i32_f64 (asm)", 1);
    run<1>("cvt_i32_f64 + add + xor", 3);
    run<2>("v_lshl_add_u32", 1);
    run<3>("v_xor_b32", 1);
    run<4>("v_pk_mad_u16", 1);
    run<6>("cmp hi + cndmask/add", 2);
    return 0;
}
