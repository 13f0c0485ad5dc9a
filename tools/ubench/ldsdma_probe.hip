// Dev probe: does global_load_lds_dwordx4 put lane l's 16 bytes at M0 + 16 l for every LDS base lstm_wp.hip uses
// (12 waves, 150 KB of LDS, bases up to 153 KB, 16-byte aligned only)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma(const float* gptr, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_off) : "memory");
}
__global__ void __launch_bounds__(768) probe(const float* src, unsigned xs_off, unsigned* bad, int stride_floats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 38000; e += 768) reinterpret_cast<unsigned*>(smem)[e] = 0xDEADBEEFu;
    __syncthreads();
    if (wave < 8) {
        for (unsigned ring = 0; ring < 4; ++ring) {
            // lane (line = lane & 15, us = lane >> 4) reads 4 floats of row `line` (rows stride_floats apart), like load_x
            const float* g = src + (size_t)(ring * 16 + (lane & 15)) * stride_floats + wave * 16 + (lane >> 4) * 4;
            dma(g, xs_off + (ring * 8u + (unsigned)wave) * 1024u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (unsigned ring = 0; ring < 4; ++ring) {
            const float4 v = *reinterpret_cast<const float4*>(smem + xs_off + (ring * 8u + wave) * 1024u + lane * 16);
            const float* g = src + (size_t)(ring * 16 + (lane & 15)) * stride_floats + wave * 16 + (lane >> 4) * 4;
            const bool ok = v.x == g[0] && v.y == g[1] && v.z == g[2] && v.w == g[3];
            if (!ok) atomicOr(&bad[(ring * 8 + wave) * 2 + (lane >> 5)], 1u << (lane & 31));
        }
    }
}
int main() {
    const int stride = 1600, rows = 64;
    std::vector<float> h((size_t)rows * stride);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 100003) + 0.25f;
    float* d; unsigned* bad;
    hipMalloc(&d, h.size() * 4); hipMalloc(&bad, 64 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
    for (unsigned xs : {22800u, 57744u, 121104u, 121088u, 98304u}) {
        hipMemset(bad, 0, 64 * 4);
        hipLaunchKernelGGL(probe, dim3(8), dim3(768), 155 * 1024, 0, d, xs, bad, stride);
        unsigned hb[64];
        hipMemcpy(hb, bad, 64 * 4, hipMemcpyDeviceToHost);
        int nbad = 0;
        for (int i = 0; i < 64; ++i) nbad += __builtin_popcount(hb[i]);
        printf("xs_off %6u: %d bad lanes of 2048", xs, nbad);
        for (int i = 0; i < 64; i += 2) if (hb[i] | hb[i + 1]) printf(" [ring %d wave %d: %08x%08x]", i / 16, (i / 2) % 8, hb[i + 1], hb[i]);
        printf("\n");
    }
    return 0;
}
