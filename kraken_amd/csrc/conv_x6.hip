// "bf16x6": convolutions in FRONT of a GroupNorm on the bf16 matrix cores (round 4).  Reference: kraken/lib/vgsl/layers.py
// ActConv2D.forward :842-860 (+ fused MaxPool :381-388); the consumer is GroupNorm.forward :967-984.
//
// Why: GroupNorm divides by the group's standard deviation and amplifies the error of its input by |x| / sigma; the 16 mantissa bits
// of the split-bf16 ("bf16x3") operands reached 2.3e-3 against the 1e-3 parity gate on random networks (profiles/r02_fuzz_300s.txt),
// so every layer up to a plan's last GroupNorm ran on the exact-f32 matrix cores -- at 1/16 of the bf16 rate (BENCH-B's second
// convolution: 1.9 of the batch's 4.1 ms).  Three bf16 pieces carry ALL 24 mantissa bits of an fp32 value
//      x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)        (|x - (h + m + l)| <= 2^-24 |x|)
// and a product keeps every term down to 2^-16:   a b ~ ah bh + ah bm + am bh + ah bl + al bh + am bm   (dropped: 2^-24 and below)
// = SIX v_mfma_f32_32x32x16_bf16 with an fp32 accumulator: an fp32-class product (~2.4e-7 relative, against the 6e-8 of one fp32
// rounding) at 6/16 of the f32 matrix cores' time.
//
// Kernel = conv_x3.hip's structure (implicit GEMM over taps x 16-channel blocks, weights ringed through LDS, two workgroups per CU)
// with three planes: input = three bf16 planes in NHWC order (split3_nhwc_kernel, norm_x3.hip), weights packed
// [chunk][tap][block][plane 3][lane][8], output = fp32 NCHW (what the GroupNorm kernels read), optional fused 2x2 pool.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ int fdiv6(int e, int d, float inv, int& rem) {
    int q = (int)((float)e * inv);
    int r = e - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

template <int POOL, int CB>
__global__ void __launch_bounds__(256, 2) conv_x6_kernel(const X3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    unsigned char* tile = smem8;                       // planes h, m, l (+ a.lds_plane bytes each)
    constexpr int SBYTES = CB * 3 * 1024;              // one (chunk, tap) record of CB filter blocks: one weight stage
    unsigned char* wring = smem8 + 3 * a.lds_plane;    // weight ring: 3 stages

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

    int vb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) vb[s] = ((srow[s] * a.sh) * a.IW + (scol[s] + px) * a.sw) * a.PSTR + half * 16;

    const int npix = a.IH * a.IW;
    const int q_per_px = a.cchunk >> 3;                  // 16-byte pieces per pixel per plane
    const int items = 3 * npix * q_per_px;
    const float inv_q = 1.0f / (float)q_per_px, inv_iw = 1.0f / (float)a.IW, inv_np = 1.0f / (float)(npix * q_per_px);
    const int gh0 = h0 * a.sh - a.ph, gw0 = w0 * a.sw - a.pw;
    const int ntaps = a.kh * a.kw;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    for (int ci = 0; ci < a.nchunks; ++ci) {
        // ---------------------------------------------------------------- stage chunk ci (16-byte copies, three planes)
        __syncthreads();   // previous chunk fully consumed
        constexpr int SB = 8;
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
                    const int plane = fdiv6(e, npix * q_per_px, inv_np, r);
                    const int pix = fdiv6(r, q_per_px, inv_q, q);
                    const int ih = fdiv6(pix, a.IW, inv_iw, iw);
                    const int gh = gh0 + ih, gw = gw0 + iw;
                    const int gc = ci * a.cchunk + q * 8;
                    dst[i] = plane * a.lds_plane + pix * a.PSTR + q * 16;
                    if (gc < a.Cin && gh >= 0 && gh < a.H && gw >= 0 && gw < len_in) {
                        const __bf16* src = a.x + (size_t)plane * a.x_plane + (((size_t)n * a.H + gh) * a.W + gw) * a.Cin + gc;
                        v[i] = *reinterpret_cast<const f32x4*>(src);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < SB; ++i)
                if (dst[i] >= 0) *reinterpret_cast<f32x4*>(tile + dst[i]) = v[i];
        }
        __syncthreads();

        // ---------------------------------------------------------------- K loop: one (tap, 16-channel block) per weight stage
        const int kbn = (ci + 1 == a.nchunks) ? a.KB_last : a.KB;
        const int nit = ntaps * kbn;
        const size_t wkb = (size_t)a.CBpad * 1536;          // elements per (tap, kb) record: CBpad x 3 planes x 512
        const __bf16* wrec0 = a.wpack + ((size_t)ci * ntaps * a.KB * a.CBpad + cb0) * 1536 + lane * 8;
        // a stage = CB x 3 KB; piece p (1 KB) -> wave p % 4
        auto issue = [&](int st, int slot) {
#pragma unroll
            for (int k = 0; k < (3 * CB + 3) / 4; ++k) {
                const int p = wave + 4 * k;
                if (p < 3 * CB)
                    __builtin_amdgcn_global_load_lds((const void*)(wrec0 + (size_t)st * wkb + p * 512), (lds_ptr)(wring + slot * SBYTES + p * 1024),
                                                     16, 0, 0);
            }
        };
        const int mp = (3 * CB - wave + 3) / 4;          // pieces p = wave + 4k < 3 CB
        int dy = 0, dx = 0, kb = 0;
        issue(0, 0);
        if (nit > 1) issue(1, 1);
        int slot = 0;
        for (int st = 0; st < nit; ++st) {
            // my copies of stage st have landed; those of stage st + 1 (mp = this wave's pieces per stage) may still fly
            if (st + 1 < nit && mp == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (st + 1 < nit && mp == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // stage st landed for everyone; stage st-1 fully read
            if (st + 2 < nit) issue(st + 2, slot >= 1 ? slot - 1 : 2);
            if (any_live) {
                const unsigned char* wst = wring + slot * SBYTES + lane * 16;
                const int xoff = (dy * a.dh * a.IW + dx * a.dw) * a.PSTR + kb * 32;
                bf16x8 xh[2], xm[2], xl[2], wh[CB], wm[CB], wl[CB];
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    xh[sg] = *reinterpret_cast<const bf16x8*>(tile + vb[sg] + xoff);
                    xm[sg] = *reinterpret_cast<const bf16x8*>(tile + a.lds_plane + vb[sg] + xoff);
                    xl[sg] = *reinterpret_cast<const bf16x8*>(tile + 2 * a.lds_plane + vb[sg] + xoff);
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    wh[cb] = *reinterpret_cast<const bf16x8*>(wst + cb * 3072);
                    wm[cb] = *reinterpret_cast<const bf16x8*>(wst + cb * 3072 + 1024);
                    wl[cb] = *reinterpret_cast<const bf16x8*>(wst + cb * 3072 + 2048);
                }
                // D[filter][pixel]; smallest terms first
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int sg = 0; sg < 2; ++sg) {
                        acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[cb], xm[sg], acc[cb][sg], 0, 0, 0);
                        acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], xh[sg], acc[cb][sg], 0, 0, 0);
                        acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xl[sg], acc[cb][sg], 0, 0, 0);
                        acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[cb], xh[sg], acc[cb][sg], 0, 0, 0);
                        acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xm[sg], acc[cb][sg], 0, 0, 0);
                        acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xh[sg], acc[cb][sg], 0, 0, 0);
                    }
            }
            if (++kb == kbn) {
                kb = 0;
                if (++dx == a.kw) { dx = 0; ++dy; }
            }
            slot = slot == 2 ? 0 : slot + 1;
        }
    }

    // ------------------------------------------------------------------------------- epilogue: fp32 NCHW (+ 2x2 max-pool)
    // lane = pixel px of its segment, register 4 rq + i = filter 8 rq + 4 half + i of the block: a register of all lanes is a run
    // of 32 (16 pooled) consecutive columns of one filter plane
    float* y = reinterpret_cast<float*>(a.y);
    auto store_tile = [&](auto actf) {
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
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = (cb0 + cb) * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                    float v = acc[cb][s][r];
                    if (POOL) {
                        v = fmaxf(v, acc[cb][1][r]);
                        v = fmaxf(v, __shfl_xor(v, 1));
                    }
                    v = actf(v + a.bias[min(co, a.CBpad * 32 - 1)]);
                    if (col >= len_out) v = 0.f;
                    if (st && co < a.Cout) y[(((size_t)n * a.Cout + co) * a.Hy + row) * a.Wy + col] = v;
                }
        }
    };
    if (a.act == ACT_RELU) store_tile([](float v) { return fmaxf(v, 0.f); });
    else store_tile([&](float v) { return krk_act(v, a.act); });
}

template <int POOL>
int launch6(const X3Args& a, int cb, dim3 grid, size_t lds, hipStream_t s) {
#define KRK_LAUNCH(CB_)                                                                         \
    do {                                                                                        \
        auto kfn = conv_x6_kernel<POOL, CB_>;                                                   \
        if (lds > 48 * 1024)                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);                                    \
    } while (0)
    switch (cb) {
        case 1: KRK_LAUNCH(1); break;
        default: KRK_LAUNCH(2); break;
    }
#undef KRK_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// filter blocks per workgroup of the three-plane kernel: at most 2 (six MFMAs per product: 128 accumulator registers would leave
// no room for the 18 fragment registers per segment pair)
int krk_x6_cb(int Cout) { return (Cout + 31) / 32 >= 2 ? 2 : 1; }

// a.x = three bf16 planes NHWC (plane stride a.x_plane), a.wpack = [chunk][tap][kb][cb][plane 3][lane][8], a.y = fp32 (N, Cout, Hy, Wy)
int krk_launch_conv_x6(const X3Args& a, bool pool, hipStream_t s) {
    const int CBt = (a.Cout + 31) / 32;
    const int cb = krk_x6_cb(a.Cout);
    dim3 grid((unsigned)(a.tiles_w * a.tiles_h * a.N), (unsigned)((CBt + cb - 1) / cb));
    const size_t lds = (size_t)3 * a.lds_plane + 3 * (size_t)cb * 3 * 1024;   // input tile (h, m, l) + weight ring
    return pool ? launch6<1>(a, cb, grid, lds, s) : launch6<0>(a, cb, grid, lds, s);
}
