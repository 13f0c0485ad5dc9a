// Microbenchmark: v_mfma_f32_32x32x2_f32 / 16x16x4 issue rate vs. independent accumulators per wave
// and waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    if (SHAPE == 32) {
        f32x16 acc[NACC];
        for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
        float s = 0; for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[NACC];
        for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
        float s = 0; for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int NACC, int SHAPE>
void run(float* d, int wgs_per_cu) {
    const int iters = 20000 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * wgs_per_cu);
    hipLaunchKernelGGL((k<NACC, SHAPE>), grid, dim3(256), 0, 0, d, 10, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SHAPE>), grid, dim3(256), 0, 0, d, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)grid.x * 4 * iters * NACC;
    double flop = mf * (SHAPE == 32 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4);
    double cyc_per_mfma_simd = ms * 1e-3 * 2.4e9 / ((double)iters * NACC * wgs_per_cu);
    printf("shape %2d acc/wave %2d waves/SIMD %d : %7.3f ms  %6.1f TFLOP/s  ~%5.1f cyc/MFMA/SIMD@2.4GHz\n", SHAPE, NACC,
           wgs_per_cu, ms, flop / ms / 1e9, cyc_per_mfma_simd);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<1, 32>(d, w); run<2, 32>(d, w); run<4, 32>(d, w); run<8, 32>(d, w);
        run<1, 16>(d, w); run<2, 16>(d, w); run<4, 16>(d, w); run<8, 16>(d, w); run<13, 16>(d, w);
    }
    return 0;
}
