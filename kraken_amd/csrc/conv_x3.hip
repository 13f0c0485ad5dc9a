// Implicit-GEMM convolution / projection on the gfx950 bf16 matrix cores with SPLIT operands
// ("bf16x3"): every fp32 value x is carried as hi = bf16(x), lo = bf16(x - hi); a product a*b is
// evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with three v_mfma_f32_32x32x16_bf16 and an fp32
// accumulator.  The dropped a_lo*b_lo term is ~2^-32 relative; the split keeps ~16 mantissa bits
// per operand, which measured against the fp32 path gives |d logit| ~ 2e-5 on BENCH-A (gate: 1e-3,
// BASELINE.json north_star) -- fp32-class results at 16/3 = 5.3x the f32 MFMA rate.
//
// Same reference call sites as conv_mfma.hip (kraken/lib/vgsl/layers.py: ActConv2D.forward :842-860,
// fused MaxPool :381-388, fused Reshape :313-335, nn.LSTM input projection :507-511, LinSoftmax :710-722).
//
// Layout: activations travel channels-last ("split NHWC"): two bf16 planes [N][H][W][C] (hi, lo) --
// the same bytes as fp32.  Channels are the contiguous K axis: one ds_read_b128 delivers the 8
// consecutive K a lane needs for a 32x32x16 MFMA.  A sequence tensor [N*T][F] is the H = 1 case.
//   tile      = (8/SR) output rows x (32*SR) output columns of one line, CB blocks of 32 filters
//   LDS       = [IH][IW] input pixels x cchunk channels per plane, pixel stride padded by 16 B
//               (conflict-free ds_read_b128 for 32/64-channel chunks)
//   staging   = straight 16-byte copies global -> LDS (the producer already wrote split NHWC)
//   K loop    = taps (dy,dx) x 16-channel blocks; weights ring through LDS in 8 KB stages (asynchronous
//               global -> LDS copies two stages ahead, one barrier per 24 MFMAs), shared by the four waves
//   epilogue  = bias + activation (+ 2x2 max-pool) + length mask, written either as split NHWC
//               (next conv / projection) or as fp32 rows [pixel][filter] (LSTM gates, logits)
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ int fdiv(int e, int d, float inv, int& rem) {
    int q = (int)((float)e * inv);
    int r = e - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

template <int POOL, int OUT_F32, int CB>
#ifndef KRK_X3_OCC
#define KRK_X3_OCC 2
#endif
__global__ void __launch_bounds__(256, KRK_X3_OCC) conv_x3_kernel(const X3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    unsigned char* tile = smem8;                       // hi plane, then lo plane (+ a.lds_plane bytes)
    unsigned char* wring = smem8 + 2 * a.lds_plane;    // weight ring: 3 stages x 8 KB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, px = lane & 31;

    int bt = blockIdx.x;
    const int tw = bt % a.tiles_w;
    bt /= a.tiles_w;
    const int th = bt % a.tiles_h;
    const int n = bt / a.tiles_h;
    const int SR = a.SR;
    const int TH = 8 / SR, TW = 32 * SR;
    const int h0 = th * TH, w0 = tw * TW;
    const int cb0 = blockIdx.y * CB;

    const int len_in = a.len_in ? a.len_in[n] : a.W;
    const int len_out = a.len_out ? a.len_out[n] : a.Wy;
    const int wlim = POOL ? min(a.Wo, 2 * len_out) : min(a.Wo, len_out);

    int srow[2], scol[2];
    bool inb[2], live[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (POOL) {
            srow[s] = 2 * (wave / SR) + s;
            scol[s] = 32 * (wave % SR);
        } else {
            const int g = wave * 2 + s;
            srow[s] = g / SR;
            scol[s] = 32 * (g % SR);
        }
        inb[s] = (h0 + srow[s] < a.Ho) && (w0 + scol[s] < a.Wo);
        live[s] = inb[s] && (w0 + scol[s] < wlim);
    }
    const bool any_live = live[0] || live[1];

    f32x16 acc[CB][2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][s][r] = 0.f;

    // byte offset of this lane's pixel (per segment) inside a plane of the LDS tile
    int vb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) vb[s] = ((srow[s] * a.sh) * a.IW + (scol[s] + px) * a.sw) * a.PSTR + half * 16;

    const int npix = a.IH * a.IW;
    const int q_per_px = a.cchunk >> 3;                  // 16-byte pieces per pixel per plane
    const int items = 2 * npix * q_per_px;
    const float inv_q = 1.0f / (float)q_per_px, inv_iw = 1.0f / (float)a.IW, inv_np = 1.0f / (float)(npix * q_per_px);
    const int gh0 = h0 * a.sh - a.ph, gw0 = w0 * a.sw - a.pw;
    const int ntaps = a.kh * a.kw;

    for (int ci = 0; ci < a.nchunks; ++ci) {
        // ---------------------------------------------------------------- stage chunk ci (16-byte copies)
        __syncthreads();   // previous chunk fully consumed
#ifndef KRK_X3_SB
#define KRK_X3_SB 8
#endif
        constexpr int SB = KRK_X3_SB;
        for (int i0 = 0; i0 * 256 < items; i0 += SB) {
            f32x4 v[SB];
            int dst[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int e = tid + 256 * (i0 + i);
                v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                dst[i] = -1;
                if (e < items) {
                    int r, q, iw;
                    const int plane = fdiv(e, npix * q_per_px, inv_np, r);
                    const int pix = fdiv(r, q_per_px, inv_q, q);
                    const int ih = fdiv(pix, a.IW, inv_iw, iw);
                    const int gh = gh0 + ih, gw = gw0 + iw;
                    const int gc = ci * a.cchunk + q * 8;
                    dst[i] = plane * a.lds_plane + pix * a.PSTR + q * 16;
                    if (gc < a.Cin && gh >= 0 && gh < a.H && gw >= 0 && gw < len_in && !KRK_DBGBIT(a, 2)) {
                        const __bf16* src = a.x + (size_t)plane * a.x_plane +
                                            (((size_t)n * a.H + gh) * a.W + gw) * a.Cin + gc;
                        v[i] = *reinterpret_cast<const f32x4*>(src);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < SB; ++i)
                if (dst[i] >= 0) *reinterpret_cast<f32x4*>(tile + dst[i]) = v[i];
        }
        __syncthreads();

        // ---------------------------------------------------------------- K loop: taps x 16-channel blocks
        // Weights travel through LDS: per-wave fragment loads from L2 need ~85 B/clk/CU at full MFMA rate against a
        // 64 B/clk L1 path, and all four waves want the same bytes.  A stage = IT iterations x CB blocks x (hi, lo) =
        // 8 KB = 24 MFMAs per wave; three stages ring through LDS, filled by global_load_lds_dwordx4 two stages ahead
        // (2 copies per wave and stage), counted vmcnt + one raw s_barrier per stage -- the gemm_x3.hip pipeline.
        const int kbn = (ci + 1 == a.nchunks) ? a.KB_last : a.KB;
        const int nit = ntaps * kbn;                      // iteration it = tap * kbn + kb: record it of this chunk
        constexpr int IT = 4 / CB;
        const int nst = (nit + IT - 1) / IT;
        const size_t wkb = (size_t)a.CBpad * 1024;        // elements per (tap, kb) record
        const __bf16* wrec0 = a.wpack + ((size_t)ci * ntaps * a.KB * a.CBpad + cb0) * 1024 + lane * 8;
        typedef __attribute__((address_space(3))) void* lds_ptr;
        auto issue = [&](int st, int slot) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int p = wave * 2 + k;               // 1 KB piece of the stage: (iteration, block, plane)
                const int it_in = p / (2 * CB), rem = p - it_in * (2 * CB);
                const __bf16* src = wrec0 + (size_t)(st * IT + it_in) * wkb + rem * 512;
                __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr)(wring + slot * 8192 + p * 1024), 16, 0, 0);
            }
        };

        int dy = 0, dx = 0, kb = 0;                       // coordinates of the iteration being computed
        issue(0, 0);
        if (nst > 1) issue(1, 1);
        int slot = 0;
        for (int st = 0; st < nst; ++st) {
            if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // stage st landed for everyone; stage st-1 fully read
            if (st + 2 < nst && !KRK_DBGBIT(a, 8)) issue(st + 2, slot >= 1 ? slot - 1 : 2);
            if (any_live && !KRK_DBGBIT(a, 1)) {
                const int nk = min(IT, nit - st * IT);
                const unsigned char* wst = wring + slot * 8192 + lane * 16;
                for (int k = 0; k < nk; ++k) {
                    const int xoff = (dy * a.dh * a.IW + dx * a.dw) * a.PSTR + kb * 32;
                    bf16x8 xh[2], xl[2], wh[CB], wl[CB];
#pragma unroll
                    for (int sg = 0; sg < 2; ++sg) {
                        xh[sg] = *reinterpret_cast<const bf16x8*>(tile + vb[sg] + xoff);
                        xl[sg] = *reinterpret_cast<const bf16x8*>(tile + a.lds_plane + vb[sg] + xoff);
                    }
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        wh[cb] = *reinterpret_cast<const bf16x8*>(wst + (k * CB + cb) * 2048);
                        wl[cb] = *reinterpret_cast<const bf16x8*>(wst + (k * CB + cb) * 2048 + 1024);
                    }
                    if (++kb == kbn) {
                        kb = 0;
                        if (++dx == a.kw) { dx = 0; ++dy; }
                    }
                    // no per-segment liveness guard: a dead segment's result is masked in the epilogue and a guard
                    // would put every MFMA into its own basic block (hazard nops, no overlap)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int sg = 0; sg < 2; ++sg) {
                            if (OUT_F32) {   // D[pixel][filter]
                                acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[sg], wh[cb], acc[cb][sg], 0, 0, 0);
                                KRK_CROSS(acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[sg], wh[cb], acc[cb][sg], 0, 0, 0);
                                          acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[sg], wl[cb], acc[cb][sg], 0, 0, 0);)
                            } else {         // D[filter][pixel]
                                acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xh[sg], acc[cb][sg], 0, 0, 0);
                                KRK_CROSS(acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xl[sg], acc[cb][sg], 0, 0, 0);
                                          acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], xh[sg], acc[cb][sg], 0, 0, 0);)
                            }
                        }
                }
            }
            slot = slot == 2 ? 0 : slot + 1;
        }
    }

    // ------------------------------------------------------------------------------- epilogues
    if (OUT_F32) {
        // fp32 rows: y[((n*Wo + col)*Ho + row)*Cout + filter]
        float* y = reinterpret_cast<float*>(a.y);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!inb[s]) continue;
            const int row = h0 + srow[s];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int co = (cb0 + cb) * 32 + px;
                if (co >= a.Cout) continue;
                const float bv = a.bias[co];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = w0 + scol[s] + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (col < a.Wo) {
                        float v = krk_act(acc[cb][s][r] + bv, a.act);
                        if (col >= len_out) v = 0.f;
                        y[(((size_t)n * a.Wo + col) * a.Ho + row) * a.Cout + co] = v;
                    }
                }
            }
        }
    } else {
        // the activation is chosen once per tile, not per element: ReLU is a single v_max
        auto store_tile = [&](auto actf) {
        // split NHWC (or split sequence rows): element index n*y_sn + row*y_sr + col*y_sc + filter
        __bf16* yh = reinterpret_cast<__bf16*>(a.y);
        __bf16* yl = yh + a.y_plane;
        constexpr int nseg = POOL ? 1 : 2;
#pragma unroll
        for (int s = 0; s < nseg; ++s) {
            if (!inb[s]) continue;
            int row, col;
            bool st;
            if (POOL) {
                row = (h0 + srow[0]) >> 1;
                col = (w0 + scol[0] + px) >> 1;
                st = !(px & 1) && row < a.Hy && col < a.Wy;
            } else {
                row = h0 + srow[s];
                col = w0 + scol[s] + px;
                st = col < a.Wo;
            }
            const size_t base = (size_t)n * a.y_sn + (size_t)row * a.y_sr + (size_t)col * a.y_sc;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int co = (cb0 + cb) * 32 + 8 * rq + 4 * half;
                    bf16x4 hv, lv;
                    f32x4 fv;   // the same four values unsplit, for a GroupNorm consumer (y_f32)
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + min(co, a.CBpad * 32 - 4));   // co % 4 == 0, padded buffer
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[cb][s][4 * rq + i];
                        if (POOL) {
                            v = fmaxf(v, acc[cb][1][4 * rq + i]);
                            v = fmaxf(v, __shfl_xor(v, 1));
                        }
                        v = actf(v + bv[i]);
                        if (col >= len_out) v = 0.f;
                        const __bf16 h = (__bf16)v;
                        hv[i] = h;
                        fv[i] = v;
                        lv[i] = (__bf16)(v - (float)h);
                    }
                    if (st && co < a.Cout && !KRK_DBGBIT(a, 4)) {
                        size_t o = base + co;
                        if (a.y_blkM > 0) {   // K-blocked sequence rows: feature f = row*Cout + co -> [f/8][line*cols + col][f%8]
                            const int f = row * a.Cout + co;
                            o = ((size_t)(f >> 3) * a.y_blkM + (size_t)n * a.y_cols + col) * 8 + (f & 7);
                        }
                        if (a.y_f32) {
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + o) = fv;
                        } else {
                            *reinterpret_cast<bf16x4*>(yh + o) = hv;
                            *reinterpret_cast<bf16x4*>(yl + o) = lv;
                        }
                    }
                }
            }
        }
        };
        if (a.act == ACT_RELU) store_tile([](float v) { return fmaxf(v, 0.f); });
        else store_tile([&](float v) { return krk_act(v, a.act); });
    }
}

template <int POOL, int OUT_F32>
int launch_cb(const X3Args& a, int cb, dim3 grid, size_t lds, hipStream_t s) {
#define KRK_LAUNCH(CB_)                                                                         \
    do {                                                                                        \
        auto kfn = conv_x3_kernel<POOL, OUT_F32, CB_>;                                          \
        if (lds > 48 * 1024)                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);                                    \
    } while (0)
    switch (cb) {
        case 1: KRK_LAUNCH(1); break;
        case 2: KRK_LAUNCH(2); break;
        default: KRK_LAUNCH(4); break;
    }
#undef KRK_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// fp32 -> split bf16 planes (hi, lo), 8 elements per thread
__global__ void __launch_bounds__(256) split_kernel(const float* __restrict__ x, __bf16* __restrict__ hi,
                                                    __bf16* __restrict__ lo, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        bf16x8 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (__bf16)a[j];
            l[j] = (__bf16)(a[j] - (float)h[j]);
            h[4 + j] = (__bf16)b[j];
            l[4 + j] = (__bf16)(b[j] - (float)h[4 + j]);
        }
        reinterpret_cast<bf16x8*>(hi)[i] = h;
        reinterpret_cast<bf16x8*>(lo)[i] = l;
    }
}

// fp32 rows [M][K] -> K-blocked split planes [ceil(K/8)][M][8]: one 16-byte piece per thread, rows fastest.  ANY = false: K % 8 == 0
// (rows are 32-byte aligned: two 16-byte loads); ANY = true: any K, element loads, the last octet zero-filled.
template <bool ANY>
__global__ void __launch_bounds__(256) split_rows_kernel(const float* __restrict__ x, __bf16* __restrict__ hi,
                                                         __bf16* __restrict__ lo, int M, int K) {
    const int K8 = (K + 7) / 8;
    const size_t total = (size_t)M * K8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int pc = (int)(i / M), row = (int)(i - (size_t)pc * M);
        f32x4 a, b;
        if constexpr (ANY) {
            const float* src = x + (size_t)row * K + pc * 8;
            const int n = K - pc * 8;                               // real elements of this octet (>= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = j < n ? src[j] : 0.f;
                b[j] = 4 + j < n ? src[4 + j] : 0.f;
            }
        } else {
            const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)row * K + pc * 8);
            a = src[0]; b = src[1];
        }
        bf16x8 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (__bf16)a[j];
            l[j] = (__bf16)(a[j] - (float)h[j]);
            h[4 + j] = (__bf16)b[j];
            l[4 + j] = (__bf16)(b[j] - (float)h[4 + j]);
        }
        reinterpret_cast<bf16x8*>(hi)[i] = h;
        reinterpret_cast<bf16x8*>(lo)[i] = l;
    }
}

}  // namespace

#ifndef KRK_BF16_ONE
// `plane`: elements between the hi and the lo plane (M x K, or more when the consumer's K is padded: capi.hip seq_kpad)
int krk_launch_split_rows(const float* x, void* hi, int M, int K, size_t plane, hipStream_t s) {
    const int K8 = (K + 7) / 8;
    if (plane < (size_t)M * K8 * 8) return -1;
    const size_t total = (size_t)M * K8;
    if (!total) return 0;
    const unsigned blocks = (unsigned)min((size_t)8192, (total + 255) / 256);
    if (K % 8)
        hipLaunchKernelGGL(split_rows_kernel<true>, dim3(blocks), dim3(256), 0, s, x, reinterpret_cast<__bf16*>(hi),
                           reinterpret_cast<__bf16*>(hi) + plane, M, K);
    else
        hipLaunchKernelGGL(split_rows_kernel<false>, dim3(blocks), dim3(256), 0, s, x, reinterpret_cast<__bf16*>(hi),
                           reinterpret_cast<__bf16*>(hi) + plane, M, K);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int krk_x3_cb(int Cout) {
    const int CB = (Cout + 31) / 32;
    return CB >= 4 ? 4 : (CB >= 2 ? 2 : 1);
}
#endif

int KRK_FN(krk_launch_conv_x3)(const X3Args& a, bool out_f32, bool pool, hipStream_t s) {
    const int CBt = (a.Cout + 31) / 32;
    const int cb = krk_x3_cb(a.Cout);
    dim3 grid((unsigned)(a.tiles_w * a.tiles_h * a.N), (unsigned)((CBt + cb - 1) / cb));
    const size_t lds = (size_t)2 * a.lds_plane + 3 * 8192;   // input tile (hi, lo) + weight ring
    if (out_f32) return pool ? -1 : launch_cb<0, 1>(a, cb, grid, lds, s);
    return pool ? launch_cb<1, 0>(a, cb, grid, lds, s) : launch_cb<0, 0>(a, cb, grid, lds, s);
}

#ifndef KRK_BF16_ONE
int krk_launch_split(const float* x, void* hi, size_t plane_elems, size_t n, hipStream_t s) {
    if (n % 8) return -1;
    const size_t n8 = n / 8;
    if (!n8) return 0;
    const unsigned blocks = (unsigned)min((size_t)4096, (n8 + 255) / 256);
    hipLaunchKernelGGL(split_kernel, dim3(blocks), dim3(256), 0, s, x, reinterpret_cast<__bf16*>(hi),
                       reinterpret_cast<__bf16*>(hi) + plane_elems, n8);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
#endif
