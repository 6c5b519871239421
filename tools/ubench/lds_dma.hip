// Check of the gfx950 LDS-DMA layout k_synth relies on: global_load_lds_dwordx4 / _dword put lane L's data at
// (M0 base) + L*16 / + L*4.  Build: hipcc --offload-arch=gfx950 -O3 lds_dma.hip -o lds_dma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
struct __attribute__((aligned(16))) Sl { uint4 a[128]; uint32_t s0[128]; uint32_t s1[128]; };
__global__ void k(const char *g, const int *idx, uint32_t *out)
{
    __shared__ Sl sl[4];
    Sl &W = sl[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const char *src = g + (size_t)idx[threadIdx.x] * 24;
    if (lane < 50) { // partially populated wave: the other lanes' slots stay untouched
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)&W.a[64], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 16),
                                         (__attribute__((address_space(3))) void *)&W.s0[64], 4, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 20),
                                         (__attribute__((address_space(3))) void *)&W.s1[64], 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint4 v = W.a[64 + lane];
    uint32_t *o = out + threadIdx.x * 6;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; o[4] = W.s0[64 + lane]; o[5] = W.s1[64 + lane];
}
int main()
{
    const int nrow = 1000, nt = 256;
    std::vector<uint32_t> rows(nrow * 6);
    for (size_t i = 0; i < rows.size(); i++) rows[i] = (uint32_t)(i * 2654435761u);
    std::vector<int> idx(nt);
    for (int i = 0; i < nt; i++) idx[i] = (i * 37 + 11) % nrow;
    char *dg; int *di; uint32_t *dout;
    (void)hipMalloc(&dg, rows.size() * 4); (void)hipMalloc(&di, nt * 4); (void)hipMalloc(&dout, nt * 24);
    (void)hipMemcpy(dg, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(di, idx.data(), nt * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dout, 0xff, nt * 24);
    k<<<1, nt>>>(dg, di, dout);
    std::vector<uint32_t> out(nt * 6);
    (void)hipMemcpy(out.data(), dout, nt * 24, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < nt; t++)
        if ((t & 63) < 50)
            for (int w = 0; w < 6; w++)
                if (out[t * 6 + w] != rows[idx[t] * 6 + w]) bad++;
    printf("lds dma layout: %s (%d mismatches)\n", bad ? "UNEXPECTED" : "ok", bad);
    return bad != 0;
}
