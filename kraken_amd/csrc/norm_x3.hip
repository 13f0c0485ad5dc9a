// Layers between the GEMM-shaped ones, on the split-bf16 ("bf16x3") activation planes: GroupNorm, stand-alone MaxPool and
// the stand-alone height collapse.  All three are memory-bound; the point of having them is that networks with
// GroupNorm (kraken's test fixtures, BENCH-B) keep running on the bf16 matrix cores end to end instead of falling back
// to the f32 plan.  Activations are two bf16 planes (hi, lo) in NHWC order; every value is used as hi + lo (fp32).
//
//   GroupNorm (reference kraken/lib/vgsl/layers.py:967-984, masked statistics :976-984): three passes over the line,
//     each split over `chunks` workgroups with fixed-order partial sums (no atomics):
//       pass 0: per-channel sums -> part[0]; pass 1: group mean, per-channel centred squares -> part[1];
//       pass 2: group mean / variance, normalise, affine, length mask, re-split.
//     A thread owns one 8-channel octet of the pixels it visits (16-byte loads, fully coalesced for any group size).
//   MaxPool (layers.py:381-388): window max of hi + lo; the winning element's (hi, lo) pair is copied, not re-split.
//   Height collapse (Reshape S1(1x0)1,3, layers.py:313-335): NHWC -> K-blocked sequence rows [h*C + c over 8][n*W + w][8].
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ void unpack8(const bf16x8& h, const bf16x8& l, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i] + (float)l[i];
}

// part layout: [pass 0|1][n][chunk][C]
template <bool XF32>
__global__ void __launch_bounds__(256) gn_x3_kernel(const void* __restrict__ xv, size_t plane, __bf16* __restrict__ y,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const int* __restrict__ lens, float* __restrict__ part,
                                                    int N, int C, int H, int W, int G, float eps, int chunks, int pass) {
    extern __shared__ float sm[];          // [256][8] reduction scratch, then mean[C], rstd[C]
    // XF32: the producing convolution wrote plain fp32 NHWC for this layer (same element order as the hi plane): dividing
    // by the group's standard deviation would amplify the 2^-17 representation error of split planes by |x| / sigma
    const __bf16* x = reinterpret_cast<const __bf16*>(xv);
    const float* xf = reinterpret_cast<const float*>(xv);
    auto load8 = [&](size_t o, float (&v)[8]) {
        if (XF32) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xf + o), b = *reinterpret_cast<const f32x4*>(xf + o + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
        } else {
            unpack8(*reinterpret_cast<const bf16x8*>(x + o), *reinterpret_cast<const bf16x8*>(x + plane + o), v);
        }
    };
    float* red = sm;
    float* mean_c = sm + 256 * 8;
    float* rstd_c = mean_c + C;
    const int n = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int Q = C >> 3;                  // octets per pixel (power of two, <= 256)
    const int q = tid & (Q - 1), p0 = tid / Q, P = 256 / Q;
    int L = lens ? lens[n] : W;
    L = min(max(L, 1), W);
    const int Cg = C / G;
    const float cnt = (float)Cg * (float)H * (float)L;
    const int npx = H * W;
    const int per = (npx + chunks - 1) / chunks;
    const int e0 = ch * per, e1 = min(npx, e0 + per);
    const size_t base = (size_t)n * npx * C;

    // group statistics of every channel, reduced once per pass by gn_x3_stats_kernel: stats[0|1][n][C]
    const float* stats = part + (size_t)2 * N * chunks * C;
    if (pass >= 1)
        for (int c = tid; c < C; c += 256) mean_c[c] = stats[(size_t)n * C + c];
    if (pass == 2)
        for (int c = tid; c < C; c += 256) rstd_c[c] = stats[(size_t)(N + n) * C + c];
    __syncthreads();

    if (pass < 2) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int e = e0 + p0; e < e1; e += P) {
            const int w = e % W;
            if (w >= L) continue;
            const size_t o = base + (size_t)e * C + q * 8;
            float v[8];
            load8(o, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = pass ? v[i] - mean_c[q * 8 + i] : v[i];
                acc[i] += pass ? d * d : d;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) red[tid * 8 + i] = acc[i];
        __syncthreads();
        // threads 0..C-1 own one channel each: sum the P pixel-lanes of its octet in a fixed order
        for (int c = tid; c < C; c += 256) {
            const int qq = c >> 3, i = c & 7;
            float s = 0.f;
            for (int k = 0; k < P; ++k) s += red[(k * Q + qq) * 8 + i];
            part[((size_t)(pass * N + n) * chunks + ch) * C + c] = s;
        }
    } else {
        for (int e = e0 + p0; e < e1; e += P) {
            const int w = e % W;
            const size_t o = base + (size_t)e * C + q * 8;
            bf16x8 hv, lv;
            if (w < L) {
                float v[8];
                load8(o, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = q * 8 + i;
                    const float r = (v[i] - mean_c[c]) * rstd_c[c] * gamma[c] + beta[c];
                    hv[i] = (__bf16)r;
                    lv[i] = (__bf16)(r - (float)hv[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { hv[i] = (__bf16)0.f; lv[i] = (__bf16)0.f; }
            }
            *reinterpret_cast<bf16x8*>(y + o) = hv;
            *reinterpret_cast<bf16x8*>(y + plane + o) = lv;
        }
    }
}

// Combines the per-chunk partial sums of one pass in a fixed order: one workgroup per (line, group), then every channel of
// the group gets the group's mean (which = 0) or 1/sqrt(var + eps) (which = 1) in stats[which][n][c].
__global__ void __launch_bounds__(256) gn_x3_stats_kernel(float* __restrict__ part, const int* __restrict__ lens, int N, int C, int H,
                                                          int W, int G, float eps, int chunks, int which) {
    __shared__ float red[256];
    const int n = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int Cg = C / G;
    int L = lens ? lens[n] : W;
    L = min(max(L, 1), W);
    const float cnt = (float)Cg * (float)H * (float)L;
    const float* src = part + (size_t)which * N * chunks * C;
    float s = 0.f;
    for (int e = tid; e < chunks * Cg; e += 256) s += src[((size_t)n * chunks + e / Cg) * C + g * Cg + e % Cg];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    float* stats = part + (size_t)2 * N * chunks * C;
    const float v = which ? 1.0f / sqrtf(red[0] / cnt + eps) : red[0] / cnt;
    for (int j = tid; j < Cg; j += 256) stats[(size_t)(which * N + n) * C + g * Cg + j] = v;
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

}  // namespace

bool krk_gn_x3_supported(int C, int G) { return C >= 8 && C <= 2048 && (C & (C - 1)) == 0 && G > 0 && C % G == 0; }

int krk_gn_x3_chunks(int N, int H, int W) {
    // memory-latency bound passes: aim at ~4 workgroups per CU (about 4096 in total), at least 512 pixels each
    const long px = (long)H * W;
    long chunks = (px + 511) / 512;
    const long cap = (4096 + N - 1) / N;
    if (chunks > cap) chunks = cap;
    return (int)(chunks < 1 ? 1 : (chunks > 1024 ? 1024 : chunks));
}

// `part`: 2 * N * chunks * C floats of partial sums + 2 * N * C floats of group statistics
int krk_launch_gn_x3(const void* x, int x_f32, void* y, size_t plane, const float* gamma, const float* beta, const int* lens,
                     float* part, int N, int C, int H, int W, int G, float eps, hipStream_t s) {
    if (!krk_gn_x3_supported(C, G)) return -4;
    const int chunks = krk_gn_x3_chunks(N, H, W);
    const size_t lds = (size_t)(256 * 8 + 2 * C) * sizeof(float);
    for (int pass = 0; pass < 3; ++pass) {
        if (pass > 0)
            hipLaunchKernelGGL(gn_x3_stats_kernel, dim3(G, N), dim3(256), 0, s, part, lens, N, C, H, W, G, eps, chunks, pass - 1);
        if (x_f32)
            hipLaunchKernelGGL(gn_x3_kernel<true>, dim3(chunks, N), dim3(256), lds, s, x, plane, (__bf16*)y, gamma, beta, lens, part, N, C,
                               H, W, G, eps, chunks, pass);
        else
            hipLaunchKernelGGL(gn_x3_kernel<false>, dim3(chunks, N), dim3(256), lds, s, x, plane, (__bf16*)y, gamma, beta, lens, part, N,
                               C, H, W, G, eps, chunks, pass);
    }
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
