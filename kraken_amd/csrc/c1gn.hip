// First convolution of a one-channel image fused INTO the GroupNorm (+ 2x2 MaxPool) behind it (round 4; exact-f32 part of a plan).
// Reference: kraken/lib/vgsl/layers.py ActConv2D.forward :842-860 (Conv2d(1 -> C, 3x3, same padding) + activation), GroupNorm.forward
// :967-984 (statistics over the line's valid width only), MaxPool.forward :381-388.
//
// Why: BENCH-B (`Cr3,3,32 Gn32 Mp2,2 ...`) wrote the first layer's 256 x 32 x 48 x 1200 fp32 output (1.9 GB) once and read it twice
// (statistics, then normalise + pool): 5.7 GB of HBM traffic and 2.1 of the batch's 5.3 ms for 9 multiply-adds per output value
// (profiles/r03_bench_b.txt).  The layer is cheap enough to be RECOMPUTED: both passes run the 3x3 filter on the vector ALUs straight
// from the 59 MB input -- the full-size tensor never exists.
//   pass 1  c1gn_stats_kernel  per (line, block of rows): per-channel sum and sum of squares of act(conv) over the valid columns
//           (fp32 per thread and wave, fp64 across waves / row blocks / the channels of a group) -> part[(n, g)][block][2]
//   pass 2  c1gn_apply_kernel  per (line, block of pooled rows): recompute the 2x2 window's four convolution values per channel,
//           normalise with the group's moments (gn_moments' arithmetic: fp64 mean / variance, float rstd), mask, max, store pooled
// The convolution is an fmaf chain in (ky, kx) order starting from the bias: bit-identical in both passes.  Exact fp32 throughout
// (GroupNorm divides by sigma: DESIGN.md section 3).
#include "common.h"

namespace {

constexpr int CMAX = 64;

__device__ __forceinline__ float c1_act(float v, int act) {
    return act == ACT_RELU ? fmaxf(v, 0.f) : krk_act(v, act);
}

// Blocking (both passes): a thread owns a column and RS (RB) consecutive rows at a time; the channel loop is OUTSIDE the row loop,
// so a channel's nine taps + bias (wave-uniform scalar loads) are fetched once per RS x 9 (RB x 36) multiply-adds, not once per
// pixel -- the first version reloaded them per pixel and ran at the scalar cache's pace.
constexpr int RS = 4;     // rows per block in the statistics pass
constexpr int RB = 3;     // (pooled) rows per workgroup in the apply pass

__global__ void __launch_bounds__(256) c1gn_stats_kernel(const C1GnArgs a) {
    __shared__ double red[4][CMAX][2];
    const int n = blockIdx.x, ch = blockIdx.y;
    const int per = (a.H + a.chunks - 1) / a.chunks, r0 = ch * per, r1 = min(a.H, r0 + per);
    int L = a.lens ? a.lens[n] : a.W;
    L = min(max(L, 1), a.W);
    const float* xn = a.x + (size_t)n * a.H * a.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (conditional loads: clamped indices + select were tried and LOST -- 0.56 -> 0.68 ms on BENCH-B; so did a wave-uniform
    // "window is interior: plain loads" fast path with padded channels -- 1.0 ms: profiles/r04_bench_b.txt)
    auto px = [&](int r, int c) -> float { return (r >= 0 && r < a.H && c >= 0 && c < L) ? xn[(size_t)r * a.W + c] : 0.f; };

    for (int c0 = 0; c0 < a.C; c0 += 16) {           // 16 channels per sweep: 32 accumulators per thread
        float s[16], q[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] = q[k] = 0.f;
        for (int col = threadIdx.x; col < L; col += 256) {
            for (int rb = r0; rb < r1; rb += RS) {
                float win[RS + 2][3];
#pragma unroll
                for (int i = 0; i < RS + 2; ++i)
#pragma unroll
                    for (int d = 0; d < 3; ++d) win[i][d] = px(rb - 1 + i, col - 1 + d);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int c = c0 + k;
                    if (c < a.C) {
                        const float* wc = a.w + c * 9;
                        const float w0 = wc[0], w1 = wc[1], w2 = wc[2], w3 = wc[3], w4 = wc[4], w5 = wc[5], w6 = wc[6], w7 = wc[7], w8 = wc[8];
                        const float bc = a.bias[c];
#pragma unroll
                        for (int i = 0; i < RS; ++i) {
                            float v = bc;
                            v = fmaf(w0, win[i][0], v); v = fmaf(w1, win[i][1], v); v = fmaf(w2, win[i][2], v);
                            v = fmaf(w3, win[i + 1][0], v); v = fmaf(w4, win[i + 1][1], v); v = fmaf(w5, win[i + 1][2], v);
                            v = fmaf(w6, win[i + 2][0], v); v = fmaf(w7, win[i + 2][1], v); v = fmaf(w8, win[i + 2][2], v);
                            v = c1_act(v, a.act);
                            v = (rb + i < r1) ? v : 0.f;
                            s[k] += v;
                            q[k] = fmaf(v, v, q[k]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s[k] += __shfl_xor(s[k], o);
                q[k] += __shfl_xor(q[k], o);
            }
            if (lane == 0 && c0 + k < a.C) {
                red[wave][c0 + k][0] = (double)s[k];
                red[wave][c0 + k][1] = (double)q[k];
            }
        }
    }
    __syncthreads();
    const int Cg = a.C / a.G;
    if (threadIdx.x < a.G) {
        const int g = threadIdx.x;
        double S = 0.0, Q = 0.0;
        for (int c = g * Cg; c < (g + 1) * Cg; ++c)
            for (int wv = 0; wv < 4; ++wv) {
                S += red[wv][c][0];
                Q += red[wv][c][1];
            }
        a.part[(((size_t)n * a.G + g) * a.chunks + ch) * 2] = S;
        a.part[(((size_t)n * a.G + g) * a.chunks + ch) * 2 + 1] = Q;
    }
}

// (pooled) rows [RB * blockIdx.y, + RB) of line n (POOL: 2x2 stride 2; else plain rows): thread = output column
template <bool POOL>
__global__ void __launch_bounds__(256) c1gn_apply_kernel(const C1GnArgs a) {
    __shared__ float rstd_s[CMAX], mean_s[CMAX];
    const int n = blockIdx.x;
    int L = a.lens ? a.lens[n] : a.W;
    L = min(max(L, 1), a.W);
    const int lo = a.len_out ? a.len_out[n] : a.Wo;
    const int Cg = a.C / a.G;
    if (threadIdx.x < a.C) {
        const int c = threadIdx.x, g = c / Cg;
        // gn_moments (misc_kernels.hip): chunk partials in index order, fp64 mean / variance, float mean and rstd
        double S = 0.0, Q = 0.0;
        for (int i = 0; i < a.chunks; ++i) {
            S += a.part[(((size_t)n * a.G + g) * a.chunks + i) * 2];
            Q += a.part[(((size_t)n * a.G + g) * a.chunks + i) * 2 + 1];
        }
        const double cnt = (double)(Cg * a.H) * (double)L;
        const double m = S / cnt;
        const double var = fmax(Q / cnt - m * m, 0.0);
        mean_s[c] = (float)m;
        rstd_s[c] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    __syncthreads();
    const int rows = POOL ? a.Ho : a.H;
    const int r0 = blockIdx.y * RB;
    const float* xn = a.x + (size_t)n * a.H * a.W;
    // (conditional loads: clamped indices + select were tried and LOST -- 0.56 -> 0.68 ms on BENCH-B; so did a wave-uniform
    // "window is interior: plain loads" fast path with padded channels -- 1.0 ms: profiles/r04_bench_b.txt)
    auto px = [&](int r, int c) -> float { return (r >= 0 && r < a.H && c >= 0 && c < L) ? xn[(size_t)r * a.W + c] : 0.f; };
    constexpr int WR = POOL ? 2 * RB + 2 : RB + 2, WC = POOL ? 4 : 3;
    for (int wo = threadIdx.x; wo < a.Wo; wo += 256) {
        // input window of the block: rows (2 r0 - 1 .. 2 (r0 + RB)) x columns (2 wo - 1 .. 2 wo + 2) with the pool
        float win[WR][WC];
        const int rb = POOL ? 2 * r0 - 1 : r0 - 1, cb = POOL ? 2 * wo - 1 : wo - 1;
#pragma unroll
        for (int i = 0; i < WR; ++i)
#pragma unroll
            for (int j = 0; j < WC; ++j) win[i][j] = px(rb + i, cb + j);
        for (int c = 0; c < a.C; ++c) {
            const float* wc = a.w + c * 9;
            const float w0 = wc[0], w1 = wc[1], w2 = wc[2], w3 = wc[3], w4 = wc[4], w5 = wc[5], w6 = wc[6], w7 = wc[7], w8 = wc[8];
            const float bc = a.bias[c], rstd = rstd_s[c], mean = mean_s[c], ga = a.gamma[c], be = a.beta[c];
            auto conv = [&](int i, int j) -> float {
                float v = bc;
                v = fmaf(w0, win[i][j], v); v = fmaf(w1, win[i][j + 1], v); v = fmaf(w2, win[i][j + 2], v);
                v = fmaf(w3, win[i + 1][j], v); v = fmaf(w4, win[i + 1][j + 1], v); v = fmaf(w5, win[i + 1][j + 2], v);
                v = fmaf(w6, win[i + 2][j], v); v = fmaf(w7, win[i + 2][j + 1], v); v = fmaf(w8, win[i + 2][j + 2], v);
                return c1_act(v, a.act);
            };
            // the stand-alone GroupNorm's arithmetic: (v - mean) * rstd * gamma + beta for columns < L, zero past it
            auto norm = [&](float v, int col) -> float { return col < L ? (v - mean) * rstd * ga + be : 0.f; };
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const int ro = r0 + k;
                float out;
                if constexpr (POOL) {
                    const int c2 = 2 * wo;
                    out = fmaxf(fmaxf(norm(conv(2 * k, 0), c2), norm(conv(2 * k, 1), c2 + 1)),
                                fmaxf(norm(conv(2 * k + 1, 0), c2), norm(conv(2 * k + 1, 1), c2 + 1)));
                    if (wo >= lo) out = 0.f;
                } else {
                    out = norm(conv(k, 0), wo);
                }
                if (ro < rows) a.y[(((size_t)n * a.C + c) * rows + ro) * a.Wo + wo] = out;
            }
        }
    }
}

}  // namespace

bool krk_c1gn_supported(int Cin, int C, int kh, int kw, int sh, int sw, int dh, int dw, int G, int pool_kh, int pool_kw, int pool_sh,
                        int pool_sw) {
    const bool pool_ok = pool_kh == 0 || (pool_kh == 2 && pool_kw == 2 && pool_sh == 2 && pool_sw == 2);
    return Cin == 1 && C >= 1 && C <= CMAX && kh == 3 && kw == 3 && sh == 1 && sw == 1 && dh == 1 && dw == 1 && G >= 1 && C % G == 0 &&
           pool_ok;
}

int krk_c1gn_chunks(int N, int H) {
    // row blocks per line in the statistics pass: enough workgroups for a small batch.  A function of the INPUT geometry only: with
    // or without the pool behind the GroupNorm the partial sums are formed in the same order, so both forms are bit-identical
    int chunks = 1;
    while (chunks * 2 * RS <= H && chunks * 2 <= 16 && (long)N * chunks < 2048) chunks *= 2;
    return chunks;
}

int krk_launch_c1gn(const C1GnArgs& a, hipStream_t s) {
    if (a.N <= 0) return 0;
    dim3 grid((unsigned)a.N, (unsigned)a.chunks);
    hipLaunchKernelGGL(c1gn_stats_kernel, grid, dim3(256), 0, s, a);
    const int rows = a.pool ? a.Ho : a.H;
    dim3 agrid((unsigned)a.N, (unsigned)((rows + RB - 1) / RB));
    if (a.pool) hipLaunchKernelGGL(c1gn_apply_kernel<true>, agrid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(c1gn_apply_kernel<false>, agrid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
