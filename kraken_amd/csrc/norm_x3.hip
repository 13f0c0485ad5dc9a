// Layers between the GEMM-shaped ones, on the split-bf16 ("bf16x3") activation planes: stand-alone MaxPool and the stand-alone
// height collapse.  Both are memory-bound.  Activations are two bf16 planes (hi, lo) in NHWC order; every value is used as
// hi + lo (fp32).  (Rounds 1-2 also had a GroupNorm on split planes here; since round 3 every layer up to a network's last
// GroupNorm runs on the exact-f32 kernels -- PlanBuilder::build in capi.hip -- and the kernel is gone.)
//
//   MaxPool (layers.py:381-388): window max of hi + lo; the winning element's (hi, lo) pair is copied, not re-split.
//   Height collapse (Reshape S1(1x0)1,3, layers.py:313-335): NHWC -> K-blocked sequence rows [h*C + c over 8][n*W + w][8].
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ void unpack8(const bf16x8& h, const bf16x8& l, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i] + (float)l[i];
}

// NHWC split planes, window kh x kw, stride sh x sw, no padding (floor); columns >= len_out[n] are written as zeros
__global__ void __launch_bounds__(256) maxpool_x3_kernel(const __bf16* __restrict__ x, size_t xplane, __bf16* __restrict__ y,
                                                         size_t yplane, const int* __restrict__ len_out, int C, int H, int W,
                                                         int kh, int kw, int sh, int sw, int Ho, int Wo, size_t total) {
    const int Q = C >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % Q);
        size_t r = i / Q;
        const int wo = (int)(r % Wo);
        r /= Wo;
        const int ho = (int)(r % Ho), n = (int)(r / Ho);
        bf16x8 bh, bl;
        float best[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bh[k] = (__bf16)0.f; bl[k] = (__bf16)0.f; }
        if (!len_out || wo < len_out[n]) {
            for (int dy = 0; dy < kh; ++dy)
                for (int dx = 0; dx < kw; ++dx) {
                    const size_t o = (((size_t)n * H + ho * sh + dy) * W + wo * sw + dx) * C + q * 8;
                    const bf16x8 h = *reinterpret_cast<const bf16x8*>(x + o), l = *reinterpret_cast<const bf16x8*>(x + xplane + o);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float v = (float)h[k] + (float)l[k];
                        if (v > best[k]) { best[k] = v; bh[k] = h[k]; bl[k] = l[k]; }
                    }
                }
        }
        const size_t oo = (((size_t)n * Ho + ho) * Wo + wo) * C + q * 8;
        *reinterpret_cast<bf16x8*>(y + oo) = bh;
        *reinterpret_cast<bf16x8*>(y + yplane + oo) = bl;
    }
}

// NHWC split planes -> K-blocked sequence rows [(h*C + c)/8][n*W + w][8] (what gemm_x3.hip streams): 16-byte copies
__global__ void __launch_bounds__(256) toseq_x3_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, size_t plane,
                                                       int N, int C, int H, int W, size_t total) {
    const int Q = C >> 3;
    const size_t rows = (size_t)N * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        // destination order (row fastest within a piece index) so that the writes are contiguous
        const size_t row = i % rows, piece = i / rows;          // piece = h*Q + q
        const int h = (int)(piece / Q), q = (int)(piece % Q);
        const int n = (int)(row / W), w = (int)(row % W);
        const size_t src = (((size_t)n * H + h) * W + w) * C + q * 8;
        const size_t dst = (piece * rows + row) * 8;
        *reinterpret_cast<bf16x8*>(y + dst) = *reinterpret_cast<const bf16x8*>(x + src);
        *reinterpret_cast<bf16x8*>(y + plane + dst) = *reinterpret_cast<const bf16x8*>(x + plane + src);
    }
}

// fp32 NCHW -> K-blocked split sequence rows (round 4): the height collapse S1(1x0)1,3 (feature = h*C + c, reference layers.py:313-335)
// and the split of the rows for the bf16x3 projection in ONE pass, where a plan's exact-f32 image part (GroupNorm networks) meets
// its split-bf16 sequence part.  Before: to_seq (fp32 rows) + split_rows = two reads and two writes of the tensor.
// A thread makes one 16-byte piece per plane: 8 channels of one (line, row, column), read as 8 dwords that are each coalesced
// along the columns of the wave's lanes.
__global__ void __launch_bounds__(256) toseq_split_f32_kernel(const float* __restrict__ x, __bf16* __restrict__ y, size_t plane,
                                                              int N, int C, int H, int W, size_t total) {
    const int Q = C >> 3;
    const size_t rows = (size_t)N * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i % rows, piece = i / rows;          // piece = h*Q + q
        const int h = (int)(piece / Q), q = (int)(piece % Q);
        const int n = (int)(row / W), w = (int)(row % W);
        const float* src = x + (((size_t)n * C + q * 8) * H + h) * W + w;
        bf16x8 hv, lv;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = src[(size_t)k * H * W];
            const __bf16 hi = (__bf16)v;
            hv[k] = hi;
            lv[k] = (__bf16)(v - (float)hi);
        }
        const size_t dst = (piece * rows + row) * 8;
        *reinterpret_cast<bf16x8*>(y + dst) = hv;
        *reinterpret_cast<bf16x8*>(y + plane + dst) = lv;
    }
}

// fp32 NCHW -> three bf16 planes NHWC, x = h + m + l (conv_x6.hip's input).  Thread = (pixel, 8-channel piece), pieces of a pixel in
// adjacent lanes: the three 16-byte stores of a wave are contiguous; the 8 reads of a thread are each coalesced along the columns.
__global__ void __launch_bounds__(256) split3_nhwc_kernel(const float* __restrict__ x, __bf16* __restrict__ y, size_t plane,
                                                          int C, int H, int W, size_t total) {
    const int Q = C >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % Q);
        const size_t pix = i / Q;                                  // (n*H + h)*W + w
        const int w = (int)(pix % W);
        const size_t nh = pix / W;
        const int h = (int)(nh % H);
        const size_t n = nh / H;
        const float* src = x + ((n * C + (size_t)q * 8) * H + h) * W + w;
        bf16x8 hv, mv, lv;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = src[(size_t)k * H * W];
            const __bf16 hi = (__bf16)v;
            const float r1 = v - (float)hi;
            const __bf16 mi = (__bf16)r1;
            hv[k] = hi;
            mv[k] = mi;
            lv[k] = (__bf16)(r1 - (float)mi);
        }
        const size_t dst = pix * C + (size_t)q * 8;
        *reinterpret_cast<bf16x8*>(y + dst) = hv;
        *reinterpret_cast<bf16x8*>(y + plane + dst) = mv;
        *reinterpret_cast<bf16x8*>(y + 2 * plane + dst) = lv;
    }
}

}  // namespace

int krk_launch_split3_nhwc(const float* x, void* y, size_t plane, int N, int C, int H, int W, hipStream_t s) {
    if (C % 8) return -4;
    const size_t total = (size_t)N * H * W * (C / 8);
    if (!total) return 0;
    const unsigned blocks = (unsigned)min((size_t)16384, (total + 255) / 256);
    hipLaunchKernelGGL(split3_nhwc_kernel, dim3(blocks), dim3(256), 0, s, x, (__bf16*)y, plane, C, H, W, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int krk_launch_toseq_split_f32(const float* x, void* y, size_t plane, int N, int C, int H, int W, hipStream_t s) {
    if (C % 8) return -4;
    const size_t total = (size_t)N * H * W * (C / 8);
    if (!total) return 0;
    const unsigned blocks = (unsigned)min((size_t)16384, (total + 255) / 256);
    hipLaunchKernelGGL(toseq_split_f32_kernel, dim3(blocks), dim3(256), 0, s, x, (__bf16*)y, plane, N, C, H, W, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int krk_launch_maxpool_x3(const void* x, size_t xplane, void* y, size_t yplane, const int* len_out, int N, int C, int H, int W,
                          int kh, int kw, int sh, int sw, int Ho, int Wo, hipStream_t s) {
    if (C % 8) return -4;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    if (!total) return 0;
    const unsigned blocks = (unsigned)min((size_t)16384, (total + 255) / 256);
    hipLaunchKernelGGL(maxpool_x3_kernel, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, xplane, (__bf16*)y, yplane, len_out, C,
                       H, W, kh, kw, sh, sw, Ho, Wo, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int krk_launch_toseq_x3(const void* x, void* y, size_t plane, int N, int C, int H, int W, hipStream_t s) {
    if (C % 8) return -4;
    const size_t total = (size_t)N * H * W * (C / 8);
    if (!total) return 0;
    const unsigned blocks = (unsigned)min((size_t)16384, (total + 255) / 256);
    hipLaunchKernelGGL(toseq_x3_kernel, dim3(blocks), dim3(256), 0, s, (const __bf16*)x, (__bf16*)y, plane, N, C, H, W, total);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
