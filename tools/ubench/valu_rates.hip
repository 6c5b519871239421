// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the
// synthesis walk is made of, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 16384
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double s, int n)
{
    __shared__ unsigned scratch[4 * 1088];
    for (int i = threadIdx.x; i < 4 * 1088; i += 256) scratch[i] = 0;
    __syncthreads();
    double a[8];
    int b[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001 + i; b[i] = threadIdx.x + i; }
    const int laddr = ((threadIdx.x & 63) * 17 + (threadIdx.x >> 6) * 1088) * 4;
    const int raddr = (((threadIdx.x * 2654435761u) >> 8) & 1023) * 4; // random table lookups
    int x0 = threadIdx.x, x1 = blockIdx.x;
    for (int it = 0; it < n; it++) {
        if ((OP >= 19 && OP <= 23) || OP == 71) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
            if (OP == 8) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(s)); }
            if (OP == 9) { asm volatile("v_floor_f64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 10) b[i] += a[(i + 1) & 7] >= s ? 1 : 0;
            if (OP == 11) { asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(b[i])); }
            if (OP == 12) { atomicAdd(&scratch[(threadIdx.x & 63) * 17 + ((b[i] + it) & 15) + (threadIdx.x >> 6) * 1088], (unsigned)i); }
            if (OP == 13) { asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 14) { unsigned long long t = (unsigned long long)(unsigned)b[i] * (unsigned)b[(i + 1) & 7] + (unsigned long long)b[(i + 2) & 7]; b[i] = (int)(t >> 20); }
            if (OP == 15) { asm volatile("v_cmp_ge_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(b[i]) : "v"(a[(i + 1) & 7]), "s"(s) : "vcc"); }
            if (OP == 16) { asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(b[i]) : "v"(a[(i + 1) & 7]), "s"(s), "v"(b[(i + 1) & 7]) : "vcc"); }
            if (OP == 19) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[i]) : "v"(laddr), "n"(i * 4)); }
            if (OP == 20) { asm volatile("ds_read_i8 %0, %1 offset:%2" : "=v"(b[i]) : "v"(laddr), "n"(i * 4)); }
            if (OP == 21) { asm volatile("ds_add_u32 %0, %1 offset:%2" : : "v"(laddr), "v"(b[i]), "n"(i * 4)); }
            if (OP == 22) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[i]) : "v"(raddr), "n"(i * 4)); }
            if (OP == 23) { asm volatile("ds_read_b32 %0, %3 offset:%4\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %1, %2\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %1, %2" : "=v"(b[i]), "+v"(x0), "+v"(x1) : "v"(laddr), "n"(i * 4)); }
            if (OP == 24) { asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %0, %1" : "+v"(x0), "+v"(x1)); }
            if (OP == 30) { asm volatile("v_fract_f64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 31) { asm volatile("v_min_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7])); }
            if (OP == 32) { asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc"); }
            if (OP == 33) { float f = __int_as_float(b[i]), g = __int_as_float(b[(i + 1) & 7]); asm volatile("v_fract_f32 %0, %1" : "=v"(f) : "v"(g)); b[i] = __float_as_int(f); }
            if (OP == 34) { float f = __int_as_float(b[i]), g = __int_as_float(b[(i + 1) & 7]); asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f) : "v"(g)); b[i] = __float_as_int(f); }
            if (OP == 35) { asm volatile("v_bfe_i32 %0, %1, 8, 8" : "=v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 36) { asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7]) : "vcc"); }
            if (OP == 37) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(b[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 38) { asm volatile("v_sub_u32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 39) { asm volatile("v_add_f64 %0, %1, -0.5" : "=v"(a[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 40) { asm volatile("v_cmp_gt_f64 vcc, |%0|, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc"); }
            if (OP == 41) { asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 42) { asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 43) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(b[(i + 1) & 7]) : "vcc"); }
            if (OP == 44) { asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 45) { asm volatile("v_rndne_f64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 46) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 47) { asm volatile("v_lshrrev_b32 %0, 27, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 48) { asm volatile("v_bfe_u32 %0, %1, 27, 5" : "=v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 49) { asm volatile("v_min_u32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 50) { asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 51) { asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 52) { asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 53) { asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 54) { float f = __int_as_float(b[i]); asm volatile("v_min3_f32 %0, |%1|, |%2|, %0" : "+v"(f) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); b[i] = __float_as_int(f); }
            if (OP == 55) { asm volatile("v_alignbit_b32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 56) { asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 57) { asm volatile("v_and_b32 %0, 0x7fffff, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 58) { asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 59) { asm volatile("v_lshl_or_b32 %0, %1, 2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 60) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(b[i]), "v"(b[(i + 1) & 7]) : "vcc"); }
            if (OP == 61) { asm volatile("v_pk_min_u16 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 62) { asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 63) { asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 64) { asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 65) { asm volatile("v_add_co_u32 %0, vcc, %1, %0\n\tv_addc_co_u32 %2, vcc, %3, %2, vcc" : "+v"(b[i]), "+v"(b[(i + 4) & 7]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7]) : "vcc"); }
            if (OP == 66) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 67) { asm volatile("v_min_f32 %0, |%1|, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 68) { asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 69) { asm volatile("v_lshl_add_u32 %0, %1, 11, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 70) { asm volatile("v_pk_add_u16 %0, %1, %0 op_sel_hi:[1,1]" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 71) { asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(b[i]) : "v"(raddr), "n"(i * 4)); }
            if (OP == 72) { asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 73) { asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 74) { asm volatile("v_or_b32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 75) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 76) { asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 77) { asm volatile("v_min_f32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 78) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 79) { asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 80) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7])); }
            if (OP == 81) { asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7])); }
            if (OP == 17) { scratch[(threadIdx.x & 63) * 17 + ((b[i] + it) & 15) + (threadIdx.x >> 6) * 1088] = (unsigned)i; }
            if (OP == 18) { b[i] += scratch[(threadIdx.x & 63) * 17 + ((b[(i+1)&7] + it) & 15) + (threadIdx.x >> 6) * 1088]; }
        }
    }
    double r = 0;
    for (int i = 0; i < 8; i++) r += a[i] + b[i];
    r += scratch[threadIdx.x] + x0 + x1;
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
    run<8>("v_fma_f64", 1);
    run<9>("v_floor_f64", 1);
    run<10>("cmp_ge_f64 + cndmask + add (C)", 1);
    run<11>("v_cvt_f64_i32", 1);
    run<12>("addr calc + ds_add_u32", 1);
    run<13>("v_pk_add_u16", 1);
    run<14>("mad_u64_u32 + shift (C)", 1);
    run<15>("v_cmp_ge_f64 + v_addc", 2);
    run<16>("v_cmp_lt_u64 + v_cndmask", 2);
    run<17>("addr calc + ds_write_b32", 1);
    run<19>("ds_read_b32 (stride 17)", 1);
    run<20>("ds_read_i8 (stride 17)", 1);
    run<21>("ds_add_u32 (stride 17)", 1);
    run<22>("ds_read_b32 (random)", 1);
    run<24>("4x v_xor", 4);
    run<23>("ds_read_b32 + 4x v_xor (per group)", 1);
    run<18>("addr calc + ds_read_b32 + add", 1);
    run<30>("v_fract_f64", 1);
    run<31>("v_min_f64", 1);
    run<32>("v_cmp_lt_f64", 1);
    run<39>("v_add_f64 (inline const)", 1);
    run<40>("v_cmp_gt_f64 |abs|", 1);
    run<45>("v_rndne_f64", 1);
    run<37>("v_cvt_f32_f64", 1);
    run<33>("v_fract_f32", 1);
    run<34>("v_fma_f32", 1);
    run<44>("v_pk_fma_f32", 1);
    run<41>("v_cvt_i32_f32", 1);
    run<35>("v_bfe_i32", 1);
    run<36>("v_cmp_lt_i32 + v_cndmask", 2);
    run<43>("v_cndmask_b32", 1);
    run<38>("v_sub_u32", 1);
    run<42>("v_and_or_b32", 1);
    run<46>("v_add_u32", 1);
    run<47>("v_lshrrev_b32 imm", 1);
    run<64>("v_lshrrev_b32 vgpr shift", 1);
    run<48>("v_bfe_u32 imm", 1);
    run<68>("v_bfe_u32 vgpr offset", 1);
    run<49>("v_min_u32", 1);
    run<50>("v_min3_u32", 1);
    run<51>("v_add3_u32", 1);
    run<52>("v_xad_u32", 1);
    run<53>("v_pk_add_f32", 1);
    run<54>("v_min3_f32 |abs|", 1);
    run<66>("v_add_f32", 1);
    run<67>("v_min_f32 |abs| (VOP3)", 1);
    run<55>("v_alignbit_b32", 1);
    run<56>("v_perm_b32", 1);
    run<57>("v_and_b32 literal", 1);
    run<58>("v_mad_u32_u24", 1);
    run<59>("v_lshl_or_b32", 1);
    run<69>("v_lshl_add_u32 imm11", 1);
    run<60>("v_cmp_lt_u32", 1);
    run<61>("v_pk_min_u16", 1);
    run<70>("v_pk_add_u16", 1);
    run<62>("v_mul_u32_u24", 1);
    run<63>("v_or3_b32", 1);
    run<72>("v_sad_u32", 1);
    run<65>("v_add_co + v_addc_co (64-bit add)", 2);
    run<71>("ds_read_u16 (random)", 1);
    run<73>("v_fma_mix_f32 (f16 src0)", 1);
    run<74>("v_or_b32", 1);
    run<75>("v_mul_f32", 1);
    run<76>("v_cvt_f32_f16", 1);
    run<77>("v_min_f32 (VOP2)", 1);
    run<78>("v_pk_mul_f32", 1);
    run<80>("v_fmac_f32", 1);
    run<81>("v_lshlrev_b32 16", 1);
    return 0;
}
