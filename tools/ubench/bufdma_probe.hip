// Dev probe (round 4): what does `buffer_load_dwordx4 ... offen lds` (raw buffer -> LDS copy) do for lanes whose offset is
// out of range -- write zeros to their LDS slot, or leave it alone?  And is `soffset` part of the bounds check?
// conv_x3p.hip stages padded convolution tiles with it and relies on "out of range = 16 zero bytes in LDS".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void probe(const float* src, int nbytes, unsigned* out, unsigned soff) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    const int lane = threadIdx.x;
    for (int e = lane; e < 1024; e += 64) smem[e] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lanes 0..31 in range (16 B each), lanes 32..47 beyond num_records, lanes 48..63 the 0x80000000 sentinel
    unsigned vo = lane < 32 ? lane * 16u : (lane < 48 ? (unsigned)nbytes + (lane - 32) * 16u : 0x80000000u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)smem, 16, vo, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = lane; e < 256; e += 64) out[e] = smem[e];
}
int main() {
    std::vector<float> h(4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i + 1.0f;
    float* d; unsigned* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (unsigned soff : {0u, 256u}) {
        hipMemset(o, 0xFF, 256 * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, 512, o, soff);   // descriptor covers 512 bytes = lanes 0..31
        unsigned ho[256];
        hipMemcpy(ho, o, 256 * 4, hipMemcpyDeviceToHost);
        int ok_in = 0, zero_oob = 0, untouched_oob = 0, other = 0, zero_far = 0, untouched_far = 0;
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < 4; ++k) {
                const unsigned v = ho[l * 4 + k];
                const float want = (float)(l * 4 + k + soff / 4) + 1.0f;
                if (l < 32) { if (v == *(const unsigned*)&want) ++ok_in; else if (v == 0) ++zero_oob; else ++other; }
                else if (l < 48) { if (v == 0) ++zero_oob; else if (v == 0xDEADBEEFu) ++untouched_oob; else ++other; }
                else { if (v == 0) ++zero_far; else if (v == 0xDEADBEEFu) ++untouched_far; else ++other; }
            }
        printf("soffset %3u: in-range dwords correct %d/128 | past num_records: zero %d untouched %d | 0x80000000: zero %d untouched %d | other %d\n",
               soff, ok_in, zero_oob, untouched_oob, zero_far, untouched_far, other);
    }
    return 0;
}
