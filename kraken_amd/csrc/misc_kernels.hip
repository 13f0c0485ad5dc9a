// HBM-bound helper kernels of the recognition path: max-pool, masked GroupNorm,
// NCHW -> time-major transpose, per-timestep softmax/argmax and the CTC run-length
// collapse.  All are plain coalesced streaming kernels with wave-shuffle reductions;
// none of them is reshaped into a GEMM.
#include "common.h"
#include <algorithm>

namespace {

// ------------------------------------------------------------------- MaxPool
// reference: MaxPool.forward kraken/lib/vgsl/layers.py:381-388 (torch.nn.MaxPool2d(k, s),
// no padding, floor).  Columns >= len_out[n] are written as zero (masked padding).
__global__ void __launch_bounds__(256) maxpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const int* __restrict__ len_out, int C, int H, int W,
                                                      int kh, int kw, int sh, int sw, int Ho, int Wo,
                                                      size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int wo = (int)(e % Wo);
        size_t r = e / Wo;
        const int ho = (int)(r % Ho);
        r /= Ho;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        float m = 0.f;
        if (!len_out || wo < len_out[n]) {
            const float* xp = x + (((size_t)n * C + c) * H + (size_t)ho * sh) * W + (size_t)wo * sw;
            m = -INFINITY;
            for (int i = 0; i < kh; ++i)
                for (int j = 0; j < kw; ++j) m = fmaxf(m, xp[(size_t)i * W + j]);
        }
        y[e] = m;
    }
}

// ----------------------------------------------------------------- GroupNorm
// reference: GroupNorm.forward kraken/lib/vgsl/layers.py:967-984: fp32, eps 1e-5, affine, statistics per (line, group) over
// (C/G, H, valid width only); positions past the valid width are zero.  HBM-bound: the activation in front of a GroupNorm is
// the widest tensor of the network (BENCH-B: 1.9 GB per 256-line batch), so the layer is two streaming passes and nothing else:
//   stats : ONE read; per (line, group) sum and sum of squares, accumulated in fp64 (full-rate on gfx950) so that
//           var = E[x^2] - mean^2 needs no second, centred pass; a group is cut into `chunks` row ranges (one workgroup each) whose
//           partial sums are combined in index order -- no atomics, results do not depend on scheduling
//   apply : one read, (x - mean) * rstd * gamma + beta, masked; a directly following MaxPool (layers.py:381-388) is taken in the
//           same pass, so the normalised full-size tensor is never written (BENCH-B: -1.9 GB write, -1.9 GB read per GroupNorm)
// One wave per row (channel, image row): lanes run along the width with 16-byte loads; no per-element integer division.
struct GnGeom {
    int C, H, W, G, chunks;
    float eps;
    int kh, kw, sh, sw, Ho, Wo;   // the fused pool (kh == 0: none)
};

__device__ __forceinline__ void gn_block_sum2(double& s, double& q, double (*red)[4]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

template <bool VEC>   // VEC: W % 4 == 0, rows are 16-byte aligned
__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, const int* __restrict__ lens,
                                                       double* __restrict__ part, const GnGeom p) {
    __shared__ double red[2][4];
    const int ng = blockIdx.x, n = ng / p.G, g = ng - n * p.G, ch = blockIdx.y;   // (line, group) in x: N * G may exceed the 65535 of y
    const int Cg = p.C / p.G, rows = Cg * p.H;
    int L = lens ? lens[n] : p.W;
    L = min(max(L, 1), p.W);   // the reference clamps to [1, W] (layers.py:982)
    const float* xg = x + ((size_t)n * p.C + (size_t)g * Cg) * p.H * p.W;
    const int per = (rows + p.chunks - 1) / p.chunks, r0 = ch * per, r1 = min(rows, r0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    auto acc = [&](float v) {
        const double d = (double)v;
        s += d;
        q = fma(d, d, q);
    };
    for (int r = r0 + wave; r < r1; r += 4) {
        const float* row = xg + (size_t)r * p.W;
        if (VEC) {
            const int L4 = L >> 2;
            // four 16-byte loads per lane in flight: clamped (not branched) indices, pinned by an empty asm (see gn_apply_kernel)
            for (int ib = lane; ib < L4; ib += 256) {
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const float4*>(row)[min(ib + 64 * k, L4 - 1)];
                asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[1].z),
                             "+v"(v[1].w), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[2].w), "+v"(v[3].x), "+v"(v[3].y),
                             "+v"(v[3].z), "+v"(v[3].w));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ib + 64 * k < L4) { acc(v[k].x); acc(v[k].y); acc(v[k].z); acc(v[k].w); }
            }
            const int w = (L4 << 2) + lane;
            if (w < L) acc(row[w]);
        } else {
#pragma unroll 4
            for (int w = lane; w < L; w += 64) acc(row[w]);
        }
    }
    gn_block_sum2(s, q, red);
    if (threadIdx.x == 0) {
        part[((size_t)ng * p.chunks + ch) * 2] = s;
        part[((size_t)ng * p.chunks + ch) * 2 + 1] = q;
    }
}

// mean and 1/sqrt(var + eps) of group `ng` from the chunk partials (every thread of the workgroup computes the same values)
__device__ __forceinline__ void gn_moments(const double* __restrict__ part, int ng, int chunks, double cnt, float eps, float& mean,
                                           float& rstd) {
    double S = 0.0, Q = 0.0;
    for (int i = 0; i < chunks; ++i) {
        S += part[((size_t)ng * chunks + i) * 2];
        Q += part[((size_t)ng * chunks + i) * 2 + 1];
    }
    const double m = S / cnt;
    const double var = fmax(Q / cnt - m * m, 0.0);
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

template <bool VEC>
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const int* __restrict__ lens, const double* __restrict__ part, const GnGeom p) {
    const int ng = blockIdx.x, n = ng / p.G, g = ng - n * p.G, ch = blockIdx.y;   // (line, group) in x: N * G may exceed the 65535 of y
    const int Cg = p.C / p.G, rows = Cg * p.H;
    int L = lens ? lens[n] : p.W;
    L = min(max(L, 1), p.W);
    float mean, rstd;
    gn_moments(part, ng, p.chunks, (double)rows * (double)L, p.eps, mean, rstd);
    const size_t base = ((size_t)n * p.C + (size_t)g * Cg) * p.H * p.W;
    const int per = (rows + p.chunks - 1) / p.chunks, r0 = ch * per, r1 = min(rows, r0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = r0 + wave; r < r1; r += 4) {
        const int c = g * Cg + r / p.H;
        const float ga = gamma[c], be = beta[c];
        const float* row = x + base + (size_t)r * p.W;
        float* out = y + base + (size_t)r * p.W;
        auto norm = [&](float v, int w) { return w < L ? (v - mean) * rstd * ga + be : 0.f; };
        if (VEC) {
            const int W4 = p.W >> 2;
            auto put = [&](const float4& v, int i) {
                const int w = i << 2;
                reinterpret_cast<float4*>(out)[i] = make_float4(norm(v.x, w), norm(v.y, w + 1), norm(v.z, w + 2), norm(v.w, w + 3));
            };
            // four loads per lane at clamped (not branched) indices, pinned by an empty asm: with a conditional use hipcc sinks a
            // load to its use and splits it per component -- one 4-byte load in flight instead of four 16-byte ones
            for (int ib = lane; ib < W4; ib += 256) {
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const float4*>(row)[min(ib + 64 * k, W4 - 1)];
                asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[1].z),
                             "+v"(v[1].w), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[2].w), "+v"(v[3].x), "+v"(v[3].y),
                             "+v"(v[3].z), "+v"(v[3].w));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ib + 64 * k < W4) put(v[k], ib + 64 * k);
            }
        } else {
#pragma unroll 4
            for (int w = lane; w < p.W; w += 64) out[w] = norm(row[w], w);
        }
    }
}

// GroupNorm + MaxPool in one pass.  The pool sees what the stand-alone GroupNorm would have written: normalised values for
// columns < len_in, zeros past it; pooled columns >= len_out are written as zeros (masked padding, as krk_launch_maxpool).
// KH > 0: the fast case -- KH x 2 windows at stride (sh, 2) over rows of W % 4 == 0 floats: one 16-byte load per window row gives
// two outputs; a lane takes two such pairs per whole trip, so 2 * KH loads are in flight per wave.  KH == 0: any window, scalar loads.
template <int KH>
__global__ void __launch_bounds__(256) gn_apply_pool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const int* __restrict__ lens, const int* __restrict__ len_out,
                                                            const double* __restrict__ part, const GnGeom p) {
    const int ng = blockIdx.x, n = ng / p.G, g = ng - n * p.G, ch = blockIdx.y;   // (line, group) in x: N * G may exceed the 65535 of y
    const int Cg = p.C / p.G;
    int L = lens ? lens[n] : p.W;
    L = min(max(L, 1), p.W);
    const int lo = len_out ? len_out[n] : p.Wo;
    float mean, rstd;
    gn_moments(part, ng, p.chunks, (double)(Cg * p.H) * (double)L, p.eps, mean, rstd);
    const int orows = Cg * p.Ho;
    const int per = (orows + p.chunks - 1) / p.chunks, r0 = ch * per, r1 = min(orows, r0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ro = r0 + wave; ro < r1; ro += 4) {
        const int cl = ro / p.Ho, ho = ro - cl * p.Ho;
        const int c = g * Cg + cl;
        const float ga = gamma[c], be = beta[c];
        const float* xin = x + (((size_t)n * p.C + c) * p.H + (size_t)ho * p.sh) * p.W;
        float* out = y + (((size_t)n * p.C + c) * p.Ho + ho) * p.Wo;
        auto norm = [&](float v, int w) { return w < L ? (v - mean) * rstd * ga + be : 0.f; };
        if constexpr (KH > 0) {
            const int Wo2 = p.Wo >> 1;
            auto pooled = [&](const float4 (&v)[KH], int j) {
                const int w = j << 2;
                float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
                for (int dy = 0; dy < KH; ++dy) {
                    m0 = fmaxf(m0, fmaxf(norm(v[dy].x, w), norm(v[dy].y, w + 1)));
                    m1 = fmaxf(m1, fmaxf(norm(v[dy].z, w + 2), norm(v[dy].w, w + 3)));
                }
                const int wo = j << 1;
                reinterpret_cast<float2*>(out)[j] = make_float2(wo < lo ? m0 : 0.f, wo + 1 < lo ? m1 : 0.f);
            };
            // two output pairs per lane and trip at clamped (not branched) indices, the 2 * KH loads pinned by an empty asm
            // (see gn_apply_kernel)
            for (int jb = lane; jb < Wo2; jb += 128) {
                float4 v[2][KH];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int dy = 0; dy < KH; ++dy)
                        v[k][dy] = reinterpret_cast<const float4*>(xin + (size_t)dy * p.W)[min(jb + 64 * k, Wo2 - 1)];
#pragma unroll
                for (int dy = 0; dy < KH; ++dy)
                    asm volatile("" : "+v"(v[0][dy].x), "+v"(v[0][dy].y), "+v"(v[0][dy].z), "+v"(v[0][dy].w), "+v"(v[1][dy].x),
                                 "+v"(v[1][dy].y), "+v"(v[1][dy].z), "+v"(v[1][dy].w));
                pooled(v[0], jb);
                if (jb + 64 < Wo2) pooled(v[1], jb + 64);
            }
        } else {
            for (int wo = lane; wo < p.Wo; wo += 64) {
                float m = 0.f;
                if (wo < lo) {
                    m = -INFINITY;
                    for (int dy = 0; dy < p.kh; ++dy)
                        for (int dx = 0; dx < p.kw; ++dx) {
                            const int w = wo * p.sw + dx;
                            m = fmaxf(m, norm(w < L ? xin[(size_t)dy * p.W + w] : 0.f, w));
                        }
                }
                out[wo] = m;
            }
        }
    }
}

// ---------------------------------------------------- NCHW -> [N][W][H*C] rows
// reference: Reshape.forward kraken/lib/vgsl/layers.py:313-335 for S1(1x0)1,3 (feature
// index h*C + c) followed by the NCHW->NWC permute of TransposedSummarizingRNN (:519) /
// LinSoftmax (:716).  32x32 LDS transpose tiles: reads coalesced along w, writes along f.
__global__ void __launch_bounds__(256) to_seq_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     int C, int H, int W) {
    __shared__ float t[32][33];
    const int n = blockIdx.z;
    const int F = C * H;
    const int f0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, w = w0 + tx;
        float v = 0.f;
        if (f < F && w < W) {
            const int h = f / C, c = f - h * C;
            v = x[(((size_t)n * C + c) * H + h) * W + w];
        }
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int w = w0 + i, f = f0 + tx;
        if (w < W && f < F) y[((size_t)n * W + w) * F + f] = t[tx][i];
    }
}

// ----------------------------------------------------- image <-> sequence rows (2-D LSTM layers)
// reference: TransposedSummarizingRNN.forward on an (N, C, H, W) image (kraken/lib/vgsl/layers.py:519-547): every
// image row (Lxx) or column (Lxy, `transpose`) is one sequence: NCHW -> [(n,h)][w][c]  or  [(n,w)][h][c], and the
// LSTM output rows go back to (N, O, H, W).  32x32 LDS tiles: reads coalesced along the source's fast axis.
// yaxis = 0: pixel index p = h*W + w;  yaxis = 1: p = w*H + h.
__global__ void __launch_bounds__(256) img2rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int C, int H, int W, int yaxis) {
    __shared__ float t[32][33];
    const int n = blockIdx.z, HW = H * W;
    const int c0 = blockIdx.y * 32, q0 = blockIdx.x * 32;     // q = h*W + w (source order)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, q = q0 + tx;
        t[i][tx] = (c < C && q < HW) ? x[((size_t)n * C + c) * HW + q] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int q = q0 + i, c = c0 + tx;
        if (q < HW && c < C) {
            const int h = q / W, w = q - h * W;
            const size_t pix = yaxis ? (size_t)w * H + h : (size_t)q;
            y[((size_t)n * HW + pix) * C + c] = t[tx][i];
        }
    }
}

// `lens` (columns-as-sequences only): output columns >= lens[n] are written as zeros (masked-padding rule).
// `last`: the rows hold T = H steps per column but only the last one is kept (summarising LSTM, layers.py:537-539):
//         the destination image has height 1.
__global__ void __launch_bounds__(256) rows2img_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int C, int H, int W, int yaxis, const int* __restrict__ lens, int last) {
    if (last) {
        // summarising LSTM: keep the last step of every sequence.  Columns as sequences: (N*W, H, C) rows -> (N, C, 1, W);
        // rows as sequences: (N*H, W, C) rows -> (N, C, H, 1).  One thread per (n, c, sequence), strided reads (small tensors)
        const int n = blockIdx.z;
        const int K = yaxis ? W : H, S = yaxis ? H : W;     // sequences per line, steps per sequence
        const size_t total = (size_t)C * K;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const int c = (int)(i / K), j = (int)(i - (size_t)c * K);
            float v = x[(((size_t)n * K + j) * S + (S - 1)) * C + c];
            if (lens && yaxis && j >= lens[n]) v = 0.f;
            y[((size_t)n * C + c) * K + j] = v;
        }
        return;
    }
    __shared__ float t[32][33];
    const int n = blockIdx.z, HW = H * W;
    const int c0 = blockIdx.y * 32, q0 = blockIdx.x * 32;     // q = destination pixel h*W + w
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int q = q0 + i, c = c0 + tx;
        float v = 0.f;
        if (q < HW && c < C) {
            const int h = q / W, w = q - h * W;
            const size_t pix = yaxis ? (size_t)w * H + h : (size_t)q;
            v = x[((size_t)n * HW + pix) * C + c];
        }
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, q = q0 + tx;
        if (c < C && q < HW) {
            float v = t[tx][i];
            if (lens && (q % W) >= lens[n]) v = 0.f;
            y[((size_t)n * C + c) * HW + q] = v;
        }
    }
}

// ----------------------------------------------------- split NHWC planes -> fp32 NCHW
// Hand-over from the bf16x3 part of a plan to layers that only exist in the f32 plan (LSTMs over image rows/columns):
// x = hi + lo per element, [n][hw][c] -> [n][c][hw] through 32x32 LDS tiles.
__global__ void __launch_bounds__(256) unsplit_kernel(const __bf16* __restrict__ x, size_t plane, float* __restrict__ y,
                                                      int C, int HW) {
    __shared__ float t[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, q0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int q = q0 + i, c = c0 + tx;
        float v = 0.f;
        if (q < HW && c < C) {
            const size_t o = ((size_t)n * HW + q) * C + c;
            v = (float)x[o] + (float)x[plane + o];
        }
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, q = q0 + tx;
        if (c < C && q < HW) y[((size_t)n * C + c) * HW + q] = t[tx][i];
    }
}

// ----------------------------------------------------- softmax / argmax per step
// reference: `(logits / T).softmax(1)` (kraken/lib/vgsl/rpred.py:226, lib/models.py:115) and
// `seq[..., :L].max(dim=0)` (kraken/lib/ctc_decoder.py:65): per (line, timestep) the maximum
// class score and its lowest index.  softmax=1: the confidence is the softmax maximum
// 1/sum(exp(z - zmax)) and the tie rule is evaluated on the probabilities (exp(..) == 1).
// Contiguous class axis: one wave per (n,t) row, 64 classes per pass, shuffle reductions.
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

__global__ void __launch_bounds__(256) rowmax_rows_kernel(const float* __restrict__ sc, long sn, long st, int C,
                                                          int T, long rows, int softmax, float temp,
                                                          float* __restrict__ probs, int* __restrict__ labels,
                                                          float* __restrict__ confs) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long n = row / T, t = row - n * T;
    const float* p = sc + n * sn + t * st;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = p[c];
        if (v > m) { m = v; mi = c; }   // ascending c: first maximum kept
    }
    wave_argmax(m, mi);
    if (mi == 0x7fffffff) mi = 0;
    if (!softmax) {
        if (lane == 0) { labels[row] = mi; confs[row] = m; }
        return;
    }
    const float zmax = m / temp;
    float sum = 0.f;
    int first1 = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float e = expf(p[c] / temp - zmax);
        sum += e;
        if (e == 1.0f && c < first1) first1 = c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        first1 = min(first1, __shfl_xor(first1, o));
    }
    if (probs) {
        float* q = probs + n * sn + t * st;
        for (int c = lane; c < C; c += 64) q[c] = expf(p[c] / temp - zmax) / sum;
    }
    if (lane == 0) { labels[row] = first1; confs[row] = 1.0f / sum; }
}

// Arbitrary strides (e.g. a contiguous (N,C,T) probability tensor handed to the decoder
// operator): one thread per (n,t), lanes run along t.
__global__ void __launch_bounds__(256) rowmax_strided_kernel(const float* __restrict__ sc, long sn, long scs,
                                                             long st, int C, int T, long rows, int softmax,
                                                             float temp, float* __restrict__ probs,
                                                             int* __restrict__ labels,
                                                             float* __restrict__ confs) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const long n = row / T, t = row - n * T;
    const float* p = sc + n * sn + t * st;
    float m = -INFINITY;
    int mi = 0;
    for (int c = 0; c < C; ++c) {
        const float v = p[(long)c * scs];
        if (v > m) { m = v; mi = c; }
    }
    if (!softmax) {
        labels[row] = mi;
        confs[row] = m;
        return;
    }
    const float zmax = m / temp;
    float sum = 0.f;
    int first1 = -1;
    for (int c = 0; c < C; ++c) {
        const float e = expf(p[(long)c * scs] / temp - zmax);
        sum += e;
        if (e == 1.0f && first1 < 0) first1 = c;
    }
    if (probs) {
        float* q = probs + n * sn + t * st;
        for (int c = 0; c < C; ++c) q[(long)c * scs] = expf(p[(long)c * scs] / temp - zmax) / sum;
    }
    labels[row] = first1 < 0 ? mi : first1;
    confs[row] = 1.0f / sum;
}

// ------------------------------------------------------------- CTC collapse
// reference: greedy_decoder kraken/lib/ctc_decoder.py:66-71 -- itertools.groupby over the
// per-step labels; every run of a non-blank (!= 0) label yields
// (label, first step, last step inclusive, max confidence in the run).
// One wave per line; run starts are found in parallel, compacted with ballot/popcount.
__global__ void __launch_bounds__(64) collapse_kernel(const int* __restrict__ labels,
                                                      const float* __restrict__ confs,
                                                      const int* __restrict__ olens, int T,
                                                      int* __restrict__ o_labels, int* __restrict__ o_starts,
                                                      int* __restrict__ o_ends, float* __restrict__ o_confs,
                                                      int* __restrict__ o_counts, int t_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* lab = reinterpret_cast<int*>(smem);
    float* cf = smem + T;
    const int n = blockIdx.x, lane = threadIdx.x;
    int L = olens ? olens[n] : T;
    L = min(max(L, 0), T);
    for (int t = lane; t < L; t += 64) {
        lab[t] = labels[(size_t)n * T + t];
        cf[t] = confs[(size_t)n * T + t];
    }
    __syncthreads();
    int count = 0;
    for (int base = 0; base < L; base += 64) {
        const int t = base + lane;
        const bool valid = t < L;
        const int l = valid ? lab[t] : 0;
        const int prev = (valid && t > 0) ? lab[t - 1] : -1;
        const bool start = valid && l != 0 && l != prev;
        const unsigned long long mask = __ballot(start);
        if (start) {
            const int idx = count + __popcll(mask & ((1ull << lane) - 1ull));
            int e = t;
            float m = cf[t];
            while (e + 1 < L && lab[e + 1] == l) {
                ++e;
                m = fmaxf(m, cf[e]);
            }
            const size_t o = (size_t)n * t_stride + idx;
            o_labels[o] = l;
            o_starts[o] = t;
            o_ends[o] = e;
            o_confs[o] = m;
        }
        count += __popcll(mask);
    }
    if (lane == 0) o_counts[n] = count;
}

}  // namespace

static inline int last_ok() { return hipGetLastError() == hipSuccess ? 0 : -2; }

int krk_launch_maxpool(const float* x, float* y, const int* len_out, int N, int C, int H, int W, int kh,
                       int kw, int sh, int sw, int Ho, int Wo, hipStream_t s) {
    const size_t total = (size_t)N * C * Ho * Wo;
    if (!total) return 0;
    const unsigned blocks = (unsigned)min((size_t)8192, (total + 255) / 256);
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, s, x, y, len_out, C, H, W, kh, kw, sh, sw,
                       Ho, Wo, total);
    return last_ok();
}

int krk_groupnorm_chunks(int N, int C, int H, int W, int G, int Ho) {
    const long per_group = (long)(C / G) * H * W;
    if (per_group < (1L << 17) || (long)N * G >= 512) return 1;     // text lines: one workgroup per (line, group)
    long chunks = (per_group + (1L << 15) - 1) >> 15;                // ~32k elements per workgroup
    const long cap = (2048 + (long)N * G - 1) / ((long)N * G);      // ~2k workgroups per pass are plenty
    chunks = chunks < cap ? chunks : (cap < 1 ? 1 : cap);
    const long rows = (long)(C / G) * (Ho > 0 ? Ho : H);             // a chunk is a range of (output) rows
    return (int)(chunks < rows ? chunks : rows);
}

int krk_launch_groupnorm(const float* x, float* y, const float* gamma, const float* beta, const int* lens, const int* len_out,
                         int N, int C, int H, int W, int G, float eps, int kh, int kw, int sh, int sw, int Ho, int Wo,
                         double* scratch, hipStream_t s) {
    if (!scratch || (size_t)N * C * H * W == 0) return scratch ? 0 : -4;
    GnGeom p;
    p.C = C; p.H = H; p.W = W; p.G = G;
    p.eps = eps;
    p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.Ho = Ho; p.Wo = Wo;
    p.chunks = krk_groupnorm_chunks(N, C, H, W, G, kh ? Ho : 0);
    const dim3 grid((unsigned)N * G, p.chunks);
    const bool vec = W % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0;
    if (vec) hipLaunchKernelGGL(gn_stats_kernel<true>, grid, dim3(256), 0, s, x, lens, scratch, p);
    else hipLaunchKernelGGL(gn_stats_kernel<false>, grid, dim3(256), 0, s, x, lens, scratch, p);
    if (!kh) {
        if (vec) hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, scratch, p);
        else hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, scratch, p);
    } else {
        const int fast = (vec && kw == 2 && sw == 2 && Wo * 2 == W && kh <= 3) ? kh : 0;
        switch (fast) {
            case 1: hipLaunchKernelGGL(gn_apply_pool_kernel<1>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, len_out, scratch, p); break;
            case 2: hipLaunchKernelGGL(gn_apply_pool_kernel<2>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, len_out, scratch, p); break;
            case 3: hipLaunchKernelGGL(gn_apply_pool_kernel<3>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, len_out, scratch, p); break;
            default: hipLaunchKernelGGL(gn_apply_pool_kernel<0>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, len_out, scratch, p);
        }
    }
    return last_ok();
}

int krk_launch_to_seq(const float* x, float* y, int N, int C, int H, int W, hipStream_t s) {
    dim3 grid((W + 31) / 32, (C * H + 31) / 32, N);
    hipLaunchKernelGGL(to_seq_kernel, grid, dim3(256), 0, s, x, y, C, H, W);
    return last_ok();
}

int krk_launch_img2rows(const float* x, float* y, int N, int C, int H, int W, int yaxis, hipStream_t s) {
    dim3 grid((H * W + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(img2rows_kernel, grid, dim3(256), 0, s, x, y, C, H, W, yaxis);
    return last_ok();
}

int krk_launch_rows2img(const float* x, float* y, int N, int C, int H, int W, int yaxis, const int* lens, int last,
                        hipStream_t s) {
    dim3 grid((H * W + 31) / 32, (C + 31) / 32, N);
    if (last) grid = dim3((unsigned)std::min<size_t>(1024, ((size_t)C * (yaxis ? W : H) + 255) / 256), 1, N);
    hipLaunchKernelGGL(rows2img_kernel, grid, dim3(256), 0, s, x, y, C, H, W, yaxis, lens, last);
    return last_ok();
}

int krk_launch_unsplit(const void* x, size_t plane, float* y, int N, int C, int H, int W, hipStream_t s) {
    dim3 grid((H * W + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(unsplit_kernel, grid, dim3(256), 0, s, (const __bf16*)x, plane, y, C, H * W);
    return last_ok();
}

namespace {
// MultiParamParallel's torch.cat(outputs, dim=1) (reference layers.py:70), one member at a time: `outer` blocks of `inner`
// contiguous floats go to offset `off` of blocks `stride` apart (NCHW: outer = N, inner = C_k*H*W; sequence rows: outer = rows,
// inner = C_k).  HBM-bound copy.
__global__ void __launch_bounds__(256) concat_kernel(const float* __restrict__ x, float* __restrict__ y, size_t inner, size_t stride,
                                                     size_t off, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t o = e / inner, j = e - o * inner;
        y[o * stride + off + j] = x[e];
    }
}

// torch.nn.Softmax(dim=1) of an NCHW tensor (ActConv2D with nl = 'm', reference layers.py:814-816, 853): one thread per pixel,
// channels HW floats apart (consecutive threads = consecutive pixels: coalesced); exp(x - max) / sum in fp32 like ATen's host
// softmax.  Columns at or beyond a line's valid width stay zero (the masked-padding rule every layer keeps).
__global__ void __launch_bounds__(256) softmax_c_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W,
                                                        const int* __restrict__ lens) {
    const int n = blockIdx.y;
    const size_t HW = (size_t)H * W;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < HW; q += (size_t)gridDim.x * 256) {
        const float* xp = x + (size_t)n * C * HW + q;
        float* yp = y + (size_t)n * C * HW + q;
        if (lens && (int)(q % W) >= lens[n]) {
            for (int c = 0; c < C; ++c) yp[(size_t)c * HW] = 0.f;
            continue;
        }
        float mx = xp[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, xp[(size_t)c * HW]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(xp[(size_t)c * HW] - mx);
        for (int c = 0; c < C; ++c) yp[(size_t)c * HW] = expf(xp[(size_t)c * HW] - mx) / sum;
    }
}

// Addition (reference layers.py:205-210): y[o][j] = sum_k x[o][k*inner + j], k = 0..nk-1 in ascending order; a block of the
// input is `in_stride` floats (>= nk*inner: what is left behind the last whole piece is dropped)
__global__ void __launch_bounds__(256) chunk_sum_kernel(const float* __restrict__ x, float* __restrict__ y, size_t inner, int nk,
                                                        size_t in_stride, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t o = e / inner, j = e - o * inner;
        const float* src = x + o * in_stride + j;
        float acc = src[0];
        for (int k = 1; k < nk; ++k) acc += src[(size_t)k * inner];
        y[e] = acc;
    }
}
}  // namespace

int krk_launch_concat(const float* x, float* y, size_t outer, size_t inner, size_t stride, size_t off, hipStream_t s) {
    const size_t total = outer * inner;
    if (!total) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(concat_kernel, dim3(blocks), dim3(256), 0, s, x, y, inner, stride, off, total);
    return last_ok();
}

int krk_launch_softmax_c(const float* x, float* y, int N, int C, int H, int W, const int* lens, hipStream_t s) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>(((size_t)H * W + 255) / 256, 4096);
    hipLaunchKernelGGL(softmax_c_kernel, dim3(blocks, (unsigned)N), dim3(256), 0, s, x, y, C, H, W, lens);
    return last_ok();
}

namespace {
// zero insertion (transposed convolution = zero insertion + convolution, capi.hip conv_transposed): HBM-bound, one thread per output
__global__ void __launch_bounds__(256) upzero_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int sh, int sw, int Ho,
                                                     int Wo, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int X = (int)(e % Wo), Y = (int)((e / Wo) % Ho);
        const size_t pl = e / ((size_t)Wo * Ho);
        const bool on = (Y % sh == 0) && (X % sw == 0);
        y[e] = on ? x[(pl * H + Y / sh) * W + X / sw] : 0.f;
    }
}
}  // namespace

int krk_launch_upzero(const float* x, float* y, size_t planes, int H, int W, int sh, int sw, int Ho, int Wo, hipStream_t s) {
    const size_t total = planes * Ho * Wo;
    if (!total) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(upzero_kernel, dim3(blocks), dim3(256), 0, s, x, y, H, W, sh, sw, Ho, Wo, total);
    return last_ok();
}

namespace {
// Reshape in general (reference layers.py:313-330: reshape, permute, reshape): the output is contiguous over the permuted 5-D
// dims `d`, an element comes from sum_i coord_i * st[i] of the input.  HBM-bound; consecutive threads write consecutive floats,
// the reads are as scattered as the permutation makes them (a layer nothing in kraken's recognisers uses: correctness, not speed)
struct Permute5 { int d[5]; size_t st[5]; };
__global__ void __launch_bounds__(256) permute5_kernel(const float* __restrict__ x, float* __restrict__ y, Permute5 q, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        size_t r = e, off = 0;
#pragma unroll
        for (int i = 4; i >= 0; --i) {
            const size_t c = r % (size_t)q.d[i];
            r /= (size_t)q.d[i];
            off += c * q.st[i];
        }
        y[e] = x[off];
    }
}
}  // namespace

int krk_launch_permute5(const float* x, float* y, const int dims[5], const size_t strides[5], hipStream_t s) {
    Permute5 q;
    size_t total = 1;
    for (int i = 0; i < 5; ++i) { q.d[i] = dims[i]; q.st[i] = strides[i]; total *= (size_t)dims[i]; }
    if (!total) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(permute5_kernel, dim3(blocks), dim3(256), 0, s, x, y, q, total);
    return last_ok();
}

int krk_launch_chunk_sum(const float* x, float* y, size_t outer, size_t inner, int nk, size_t in_stride, hipStream_t s) {
    const size_t total = outer * inner;
    if (!total) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(chunk_sum_kernel, dim3(blocks), dim3(256), 0, s, x, y, inner, nk, in_stride, total);
    return last_ok();
}

namespace {
// Heatmap tail of the segmenter (reference kraken/lib/vgsl/spred.py:268-272: F.interpolate(o, size) -- nearest -- then sigmoid):
// (C, h, w) logits -> (C, H, W) probabilities in one pass.  Source index = min(floor(dst * (float)in / out), in - 1), in fp32
// like ATen's nearest_idx, so that the same pixel is picked.  HBM-bound on the (C, H, W) fp32 write.
__global__ void __launch_bounds__(256) upsample_sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int h, int w, int H, int W) {
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const size_t total = (size_t)C * H * W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int X = (int)(e % W), Y = (int)((e / W) % H), c = (int)(e / ((size_t)W * H));
        const int ys = min((int)floorf(Y * sy), h - 1), xs = min((int)floorf(X * sx), w - 1);
        const float v = x[((size_t)c * h + ys) * w + xs];
        y[e] = 1.0f / (1.0f + expf(-v));
    }
}
}  // namespace

int krk_launch_upsample_sigmoid(const float* x, float* y, int C, int h, int w, int H, int W, hipStream_t s) {
    const size_t total = (size_t)C * H * W;
    if (!total) return 0;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(upsample_sigmoid_kernel, dim3(blocks), dim3(256), 0, s, x, y, C, h, w, H, W);
    return last_ok();
}

int krk_launch_rowmax(const float* scores, long sn, long sc, long st, int N, int C, int T, int softmax,
                      float temp, float* probs, int* labels, float* confs, hipStream_t s) {
    const long rows = (long)N * T;
    if (!rows) return 0;
    if (sc == 1) {
        hipLaunchKernelGGL(rowmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, scores, sn, st,
                           C, T, rows, softmax, temp, probs, labels, confs);
    } else {
        hipLaunchKernelGGL(rowmax_strided_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, scores,
                           sn, sc, st, C, T, rows, softmax, temp, probs, labels, confs);
    }
    return last_ok();
}

int krk_launch_collapse(const int* labels, const float* confs, const int* olens, int N, int T, int* o_labels,
                        int* o_starts, int* o_ends, float* o_confs, int* o_counts, int t_stride,
                        hipStream_t s) {
    if (!N) return 0;
    const size_t lds = (size_t)2 * T * sizeof(float);
    hipLaunchKernelGGL(collapse_kernel, dim3(N), dim3(64), lds, s, labels, confs, olens, T, o_labels, o_starts,
                       o_ends, o_confs, o_counts, t_stride);
    return last_ok();
}
