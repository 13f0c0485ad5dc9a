// Dev probe: which XCD does block b of a 1-D grid run on (s_getreg HW_REG_XCC_ID), for 768-thread blocks with 150 KB of LDS
// (the geometry of lstm_wp.hip), and how long does a same-XCD vs cross-XCD granule hand-off take?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(768) who(unsigned* out, unsigned* order, unsigned* ctr) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x;
        order[blockIdx.x] = atomicAdd(ctr, 1u);
        smem[0] = 1;
    }
}
int main() {
    const int grids[] = {16, 64, 120, 256, 300};
    unsigned *d, *o, *c;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 4096 * 4); hipMalloc(&c, 4);
    hipFuncSetAttribute((const void*)who, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int g : grids) {
        hipMemset(c, 0, 4);
        hipLaunchKernelGGL(who, dim3(g), dim3(768), 150 * 1024, 0, d, o, c);
        std::vector<unsigned> h(g), ord(g);
        hipMemcpy(h.data(), d, g * 4, hipMemcpyDeviceToHost);
        hipMemcpy(ord.data(), o, g * 4, hipMemcpyDeviceToHost);
        int cnt[16] = {0}, rr = 0;
        for (int b = 0; b < g; ++b) { cnt[h[b] & 15]++; rr += ((h[b] & 15) == (unsigned)(b % 8)); }
        printf("grid %d: per-XCD counts", g);
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf(" | blocks with xcc == b%%8: %d/%d | first 24 xcc:", rr, g);
        for (int b = 0; b < 24 && b < g; ++b) printf(" %u", h[b] & 15);
        printf(" | start order of first 16 blocks:");
        for (int b = 0; b < 16 && b < g; ++b) printf(" %u", ord[b]);
        printf("\n");
    }
    return 0;
}
