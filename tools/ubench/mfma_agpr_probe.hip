// Dev probe: can ONE wave per SIMD keep the matrix pipe busy, with the A operand in AGPRs, and does its own VALU work overlap?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_agpr_probe.hip -o tools/ubench/mfma_agpr_probe && tools/ubench/mfma_agpr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool AG> __device__ __forceinline__ void mfma_aa(f32x4& acc, const u32x4& w, const bf16x8& h) {   // accumulator in the AGPR half too
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(h));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(h));
}
template <bool AG> __device__ __forceinline__ void mfma(f32x4& acc, const u32x4& w, const bf16x8& h) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(h));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(h));
}
// MODE 0: 84 MFMAs per iteration, A in AGPR (28 distinct fragments); 1: A in VGPR (4 distinct); 2: AGPR + 3 VALU between triplets;
// 3: AGPR + 9 VALU between triplets; 4: only the VALU of mode 3; 5: AGPR + 2 transcendentals between triplets
template <int MODE>
__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ wp, float* out, int iters, unsigned long long* cyc) {
    constexpr bool AG = MODE != 1 && MODE != 7;
    constexpr int NW = AG ? 28 : 4;
    u32x4 w[NW];
    const int lane = threadIdx.x & 63;
    if constexpr (AG) {
#pragma unroll
        for (int i = 0; i < NW; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(w[i]) : "v"(wp + i * 64 + lane) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+a"(w[0]), "+a"(w[NW - 1]) : : "memory");
    } else {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = wp[i * 64 + lane];
    }
    bf16x8 h[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) h[i] = __builtin_bit_cast(bf16x8, wp[(40 + i) * 64 + lane]);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
    float v0 = lane, v1 = lane * 2.f, v2 = 1.f, v3 = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kb = 0; kb < 7; ++kb) {
                if constexpr (MODE >= 8) {
                    // fine-grained: behind EVERY MFMA 3 independent VALU (8) / 3 dependent VALU (9) / a transcendental and a dependent VALU (10) / 2 dependent (11)
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3) {
                        if (t3 == 0) mfma<AG>(a0, w[(i * 7 + kb) % NW], h[kb]);
                        else if (t3 == 1) mfma<AG>(a1, w[(i * 7 + kb) % NW], h[(kb + 1) % 7]);
                        else mfma<AG>(a2, w[(i * 7 + kb + 1) % NW], h[kb]);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (MODE == 8) { v0 = __builtin_fmaf(v0, v2, v3); v1 = __builtin_fmaf(v1, v3, v2); v2 = __builtin_fmaf(v2, 0.999f, 0.001f); }
                        if constexpr (MODE == 9) { v0 = __builtin_fmaf(v0, v2, v3); v0 = __builtin_fmaf(v0, v3, v2); v0 = __builtin_fmaf(v0, 0.999f, 0.001f); }
                        if constexpr (MODE == 10) { v0 = __builtin_amdgcn_exp2f(v0); v0 = __builtin_fmaf(v0, 0.999f, 0.001f); }
                        if constexpr (MODE == 11) { v0 = __builtin_fmaf(v0, v2, v3); v0 = __builtin_fmaf(v0, v3, v2); }
                        if constexpr (MODE == 12) {      // two reads of the accumulator file behind every MFMA (A operands and accumulators live there)
                            unsigned r0, r1;
                            asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(r0), "=v"(r1) : "a"(w[(i * 7 + kb + 3) % NW][0]), "a"(w[(i * 7 + kb + 5) % NW][1]));
                            v3 = __builtin_bit_cast(float, r0 ^ r1);
                        }
                        if constexpr (MODE == 13) {      // four accumulator-file reads behind every MFMA
                            unsigned r0, r1, r2, r3;
                            asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
                                         : "a"(w[(i * 7 + kb + 3) % NW][0]), "a"(w[(i * 7 + kb + 5) % NW][1]), "a"(w[(i * 7 + kb + 7) % NW][2]), "a"(w[(i * 7 + kb + 9) % NW][3]));
                            v3 = __builtin_bit_cast(float, r0 ^ r1 ^ r2 ^ r3);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if constexpr (MODE == 6 || MODE == 7) {
                    mfma_aa<AG>(a0, w[(i * 7 + kb) % NW], h[kb]);
                    mfma_aa<AG>(a1, w[(i * 7 + kb) % NW], h[(kb + 1) % 7]);
                    mfma_aa<AG>(a2, w[(i * 7 + kb + 1) % NW], h[kb]);
                } else if constexpr (MODE != 4) {
                    mfma<AG>(a0, w[(i * 7 + kb) % NW], h[kb]);
                    mfma<AG>(a1, w[(i * 7 + kb) % NW], h[(kb + 1) % 7]);
                    mfma<AG>(a2, w[(i * 7 + kb + 1) % NW], h[kb]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == 2 || MODE == 3 || MODE == 4) {
#pragma unroll
                    for (int r = 0; r < (MODE == 2 ? 1 : 3); ++r) {
                        v0 = __builtin_fmaf(v0, v2, v3);
                        v1 = __builtin_fmaf(v1, v3, v2);
                        v2 = __builtin_fmaf(v2, 0.999f, 0.001f);
                    }
                }
                if constexpr (MODE == 5) {
                    v0 = __builtin_amdgcn_exp2f(v0);
                    v1 = __builtin_amdgcn_rcpf(v1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    if constexpr (MODE == 6 || MODE == 7) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(a0), "+a"(a1), "+a"(a2));
    else asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2));
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + v0 + v1 + v2;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
    u32x4* wp; float* out; unsigned long long* cyc;
    hipMalloc(&wp, 64 * 64 * 16); hipMemset(wp, 0, 64 * 64 * 16);
    hipMalloc(&out, 256 * 256 * 4);
    hipHostMalloc(&cyc, 128);
    const int iters = 2000;
    const char* names[] = {"84 MFMA, A in AGPR", "84 MFMA, A in VGPR", "AGPR + 3 VALU per triplet", "AGPR + 9 VALU per triplet", "9 VALU per 'triplet' alone", "AGPR + 2 transcendentals per triplet", "A in AGPR, accumulators in AGPR", "A in VGPR, accumulators in AGPR", "every MFMA + 3 independent VALU", "every MFMA + 3 DEPENDENT VALU", "every MFMA + exp + dependent VALU", "every MFMA + 2 DEPENDENT VALU", "every MFMA + 2 v_accvgpr_read", "every MFMA + 4 v_accvgpr_read"};
    for (int rep = 0; rep < 2; ++rep) {
        k<0><<<256, 256>>>(wp, out, iters, cyc); k<1><<<256, 256>>>(wp, out, iters, cyc); k<2><<<256, 256>>>(wp, out, iters, cyc);
        k<3><<<256, 256>>>(wp, out, iters, cyc); k<4><<<256, 256>>>(wp, out, iters, cyc); k<5><<<256, 256>>>(wp, out, iters, cyc); k<6><<<256, 256>>>(wp, out, iters, cyc); k<7><<<256, 256>>>(wp, out, iters, cyc); k<8><<<256, 256>>>(wp, out, iters, cyc); k<9><<<256, 256>>>(wp, out, iters, cyc); k<10><<<256, 256>>>(wp, out, iters, cyc); k<11><<<256, 256>>>(wp, out, iters, cyc); k<12><<<256, 256>>>(wp, out, iters, cyc); k<13><<<256, 256>>>(wp, out, iters, cyc);
        hipDeviceSynchronize();
    }
    for (int m = 0; m < 14; ++m) printf("%-40s %8.1f s_memtime ticks per 84-MFMA iteration (x(sclk/100MHz) = cycles)\n", names[m], (double)cyc[m] / iters);
    return 0;
}
