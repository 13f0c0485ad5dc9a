// Which CUs does a CU-masked stream use?  Every workgroup records (XCC id, SE id, CU id) and spins ~20 us.
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void where_kernel(unsigned* out, long long spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
extern "C" int where_launch(void* stream, unsigned* out, int n, long long spin) {
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, out, spin);
    return (int)hipGetLastError();
}
