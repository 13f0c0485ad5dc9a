// Microbenchmark: f32 MFMA 32x32x2 with the ingredients of the conv inner loop added one by one.
// EXTRA bits: 1 = one ds_read_b32 operand per MFMA, 2 = LDS address comes from an offset table (ds_read_b128 per 4
// K-steps + v_add per read), 4 = weights via global dwordx4 per 4 K-steps (L2 resident), 8 = sched_barrier pinning
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NSEG, int EXTRA>
__global__ void __launch_bounds__(256, 2) k(float* out, const float* __restrict__ wts, int groups) {
    extern __shared__ float lds[];
    int* otab = (int*)lds;                 // 4096 ints
    float* tile = lds + 4096;              // 8192 floats
    for (int i = threadIdx.x; i < 4096; i += 256) otab[i] = (i * 37) & 4095;
    for (int i = threadIdx.x; i < 8192; i += 256) tile[i] = i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[NSEG];
    for (int j = 0; j < NSEG; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* wp = wts + lane * 4 + (threadIdx.x >> 6) * 4096;
    int boff[NSEG];
    for (int j = 0; j < NSEG; ++j) boff[j] = (lane & 31) + 64 * j;
    f32x4 w = {1.f, 2.f, 3.f, 4.f};
    for (int g = 0; g < groups; ++g) {
        int4 o = {0, 1, 2, 3};
        if (EXTRA & 2) o = *reinterpret_cast<const int4*>(otab + 4 * (g & 1023));
        if (EXTRA & 4) w = *reinterpret_cast<const f32x4*>(wp + (size_t)(g & 255) * 256 * 16);
        float b[NSEG][4];
#pragma unroll
        for (int j = 0; j < NSEG; ++j) {
            if (EXTRA & 1) {
                b[j][0] = tile[o.x + boff[j]]; b[j][1] = tile[o.y + boff[j]];
                b[j][2] = tile[o.z + boff[j]]; b[j][3] = tile[o.w + boff[j]];
            } else { b[j][0] = 1.f + j; b[j][1] = 2.f; b[j][2] = 3.f; b[j][3] = 4.f; }
        }
        if (EXTRA & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NSEG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[e], b[j][e], acc[j], 0, 0, 0);
    }
    float s = 0; for (int j = 0; j < NSEG; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NSEG, int EXTRA>
void run(float* d, const float* w, int wgs_per_cu) {
    const int groups = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * wgs_per_cu);
    size_t lds = (4096 + 8192) * 4;
    hipLaunchKernelGGL((k<NSEG, EXTRA>), grid, dim3(256), lds, 0, d, w, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NSEG, EXTRA>), grid, dim3(256), lds, 0, d, w, groups);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)grid.x * 4 * groups * 4 * NSEG;
    printf("nseg %d extra %2d waves/SIMD %d : %7.3f ms  %6.1f TFLOP/s\n", NSEG, EXTRA, wgs_per_cu, ms, mf * 4096.0 / ms / 1e9);
}

int main() {
    float *d, *w; hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20);
    for (int wv = 1; wv <= 2; ++wv) {
        run<2, 0>(d, w, wv); run<2, 1>(d, w, wv); run<2, 3>(d, w, wv); run<2, 4>(d, w, wv); run<2, 7>(d, w, wv); run<2, 15>(d, w, wv);
        run<4, 7>(d, w, wv);
    }
    return 0;
}
