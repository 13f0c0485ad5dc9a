// Pipelined variant of conv_x3.hip (round 4): the same implicit-GEMM convolution on the bf16 matrix cores with split operands
// (reference kraken/lib/vgsl/layers.py: ActConv2D.forward :842-860, fused MaxPool :381-388, fused Reshape :313-335), the same
// tiles, weights and results -- but the input tile is staged ASYNCHRONOUSLY and the whole K loop is one flat pipeline.
//
// conv_x3.hip ran staging, K loop and epilogue of a tile strictly one after the other: per 16-channel chunk two workgroup
// barriers around a staging pass whose index arithmetic (three divisions per 16-byte piece) alone cost 0.09 / 0.065 ms of the
// 0.30 / 0.19 ms of BENCH-A's two convolutions (profiles/r03_x3_ablation.txt), and a restart of the weight ring.  Here
//   * lane l of pixel block b owns tile pixel 64 b + l (its global offset: a reciprocal multiply, no division); a chunk's tile is
//     NB <= 8 raw-buffer -> LDS copies per wave (buffer_load_dwordx4 ... lds: no staging
//     registers, no ds_write pass; out-of-range lanes -- image border, beyond the line's valid width, pixel padding -- deliver
//     zeros, probed by tools/ubench/bufdma_probe), the channel chunk / plane / 8-channel piece chosen by descriptor and soffset;
//   * LDS tile = [buffer 2][plane hi|lo][piece 2][pixel][16 B]: a fragment read (32 consecutive pixels x 16 B) is one contiguous
//     512-byte run; two buffers: chunk c + 1 lands while the MFMAs of chunk c run;
//   * the weight ring of conv_x3.hip (8 KB stages, copies two stages ahead, one raw barrier per stage) runs straight THROUGH the
//     chunk boundaries (one record per (chunk, tap): the stages are numbered over the whole tile), and the tile copies ride on its
//     stage boundaries, at most TPS per boundary: every `s_waitcnt vmcnt(n)` is "all but the group issued at the last boundary".
// Eligibility (krk_conv_x3p_supported): 16-channel chunks, <= 512 tile pixels, >= 4 weight stages per chunk; else conv_x3.hip.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kOOBp = 0x80000000u;    // voffset beyond every descriptor used here: the copy delivers zeros
constexpr int NBMAX = 8;                   // pixel blocks of 64 per tile (host-side eligibility)

template <int N>
__device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void vmwait_n(int n) {   // n is wave-uniform
    switch (n) {
        case 0: vmwait<0>(); break;
        case 1: vmwait<1>(); break;
        case 2: vmwait<2>(); break;
        case 3: vmwait<3>(); break;
        case 4: vmwait<4>(); break;
        case 5: vmwait<5>(); break;
        default: vmwait<6>(); break;
    }
}

#if defined(KRK_ABLATE) && !defined(KRK_BF16_ONE)
// phase cycles summed over waves: 0 prologue, 1 copy waits (vmcnt), 2 barrier, 3 copy issue, 4 fragment reads + MFMAs, 5 epilogue; [6] = waves
__device__ unsigned long long g_x3p_phases[8];
#endif

// SB (round 6): ONE tile buffer instead of two -- 4 x NB + 24 KB of LDS (52 KB for BENCH-A's two layers) so that THREE workgroups share a
// CU where two did (80 KB each): the next chunk's tile is copied at the chunk boundary itself (barrier, NB copies per wave, wait, barrier)
// and the other two workgroups' MFMAs cover that wait.  98-112 registers per wave: twelve waves fit.
template <int POOL, int CB, bool SB>
__global__ void __launch_bounds__(256, 2) conv_x3p_kernel(const X3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    KRK_PHASES(6);
    KRK_PH_START(a);

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

    const int npix = a.IH * a.IW;
    const int NB = (npix + 63) >> 6;
    const int PPL = NB * 1024;                           // bytes per (plane, piece) of a tile buffer
    constexpr int NTB = SB ? 1 : 2;                      // tile buffers
    unsigned char* wring = smem8 + NTB * 4 * PPL;        // weight ring: 3 stages x 8 KB

    // ---- tile copies.  Lane l of pixel block b owns tile pixel p = 64 b + l; its global byte offset inside the line is recomputed
    // per copy (a register table indexed by the run-time block number ends up in scratch memory): ih = p / IW by a 16-bit
    // reciprocal (exact for p < 512, IW < 128: krk_conv_x3p_tps), ~10 VALU per copy against 1 KB moved
    const int gh0 = h0 * a.sh - a.ph, gw0 = w0 * a.sw - a.pw;
    const unsigned iw_rcp = (65536u + (unsigned)a.IW - 1u) / (unsigned)a.IW;
    const bool no_stage = KRK_DBGBIT(a, 2);
    // wave w copies (plane w >> 1, piece w & 1) of every chunk
    const size_t line_elems = (size_t)a.H * a.W * a.Cin;
    const __bf16* lbase = a.x + (size_t)n * line_elems + (size_t)(wave >> 1) * a.x_plane;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)lbase, 0, (int)(line_elems * 2), 0x00020000);
    const int my_pp = wave * PPL;
    const int q16 = (wave & 1) * 16;
    // copy pixel block b of chunk c into tile buffer c & 1
    auto tile_copy = [&](int c, int b) {
        unsigned char* dst = smem8 + (SB ? 0 : (c & 1)) * 4 * PPL + my_pp + b * 1024;
        const unsigned so = (unsigned)(c * 32 + q16);
        const int p = b * 64 + lane;
        const int ih = (int)(((unsigned)p * iw_rcp) >> 16), iw = p - ih * a.IW;
        const int gh = gh0 + ih, gw = gw0 + iw;
        const bool ok = p < npix && gh >= 0 && gh < a.H && gw >= 0 && gw < len_in && !no_stage;
        const unsigned vo = ok ? (unsigned)((gh * a.W + gw) * a.Cin) * 2u : kOOBp;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr)dst, 16, vo, so, 0, 0);
    };

    f32x16 acc[CB][2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][s][r] = 0.f;

    // byte offset of this lane's pixel (per segment) inside the hi plane of a tile buffer
    int vb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) vb[s] = ((srow[s] * a.sh) * a.IW + (scol[s] + px) * a.sw) * 16 + half * PPL;

    const int ntaps = a.kh * a.kw;
    const int G = a.nchunks * ntaps;                     // iterations = (chunk, tap) records
    constexpr int IT = 4 / CB;                           // iterations per 8 KB weight stage
    const int NST = (G + IT - 1) / IT;
    const int TPS = a.tps;                               // tile copies per stage boundary (host: ceil(NB / (stages per chunk - 3)))
    const size_t wkb = (size_t)a.CBpad * 1024;           // elements per record
    const __bf16* wrec0 = a.wpack + (size_t)cb0 * 1024 + lane * 8;
    auto issue_w = [&](int st, int slot) {
        if (KRK_DBGBIT(a, 8)) return;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = wave * 2 + k;                  // 1 KB piece of the stage: (iteration, block, plane)
            const int it_in = p / (2 * CB), rem = p - it_in * (2 * CB);
            const __bf16* src = wrec0 + (size_t)(st * IT + it_in) * wkb + rem * 512;
            __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr)(wring + slot * 8192 + p * 1024), 16, 0, 0);
        }
    };

    // ---- prologue: the whole first chunk, weight stages 0 and 1
    for (int b = 0; b < NB; ++b) tile_copy(0, b);
    issue_w(0, 0);
    if (NST > 1) issue_w(1, 1);
    int last_cnt = KRK_DBGBIT(a, 8) ? 0 : (NST > 1 ? 2 : 0);   // copies issued after the ones the next boundary needs
    int tc = 1, tb = 0;                                  // next tile copy: chunk tc, pixel block tb
    int dy = 0, dx = 0, ci = 0;                          // coordinates of the iteration being computed
    int slot = 0;
    KRK_PH(a, 0);
    for (int st = 0; st < NST; ++st) {
        vmwait_n(last_cnt);
        KRK_PH(a, 1);
        __builtin_amdgcn_s_barrier();                    // stage st and every older copy landed for everyone; stage st-1 fully read
        KRK_PH(a, 2);
        int cnt = 0;
        if (st + 2 < NST) {
            issue_w(st + 2, slot >= 1 ? slot - 1 : 2);
            cnt = KRK_DBGBIT(a, 8) ? 0 : 2;
        }
        // tile copies of chunk tc (into the buffer chunk tc - 2 used): legal once every wave is past chunk tc - 2, i.e. from the
        // first stage that starts inside chunk tc - 1
        if (!SB && tc < a.nchunks && st * IT >= (tc - 1) * ntaps) {
            for (int k = 0; k < TPS && tb < NB; ++k, ++tb, ++cnt) tile_copy(tc, tb);
            if (tb == NB) { tb = 0; ++tc; }
        }
        last_cnt = cnt;
        KRK_PH(a, 3);
        {
            const int nk = min(IT, G - st * IT);
            const unsigned char* wst = wring + slot * 8192 + lane * 16;
            for (int k = 0; k < nk; ++k) {
                if (SB && dx == 0 && dy == 0 && ci > 0) {
                    // single tile buffer: chunk ci starts here.  Everyone has read chunk ci - 1 (their MFMAs are issued: the fragment
                    // reads have returned), so it can be overwritten; whatever else is in flight (weight stages) lands with it
                    __builtin_amdgcn_s_barrier();
                    for (int b = 0; b < NB; ++b) tile_copy(ci, b);
                    vmwait<0>();
                    __builtin_amdgcn_s_barrier();
                }
                if (!any_live || KRK_DBGBIT(a, 1)) {      // (the boundary above is for every wave; the arithmetic only for live ones)
                    if (++dx == a.kw) {
                        dx = 0;
                        if (++dy == a.kh) { dy = 0; ++ci; }
                    }
                    continue;
                }
                const unsigned char* xt = smem8 + (SB ? 0 : (ci & 1)) * 4 * PPL + (dy * a.dh * a.IW + dx * a.dw) * 16;
                bf16x8 xh[2], xl[2], wh[CB], wl[CB];
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    xh[sg] = *reinterpret_cast<const bf16x8*>(xt + vb[sg]);
                    xl[sg] = *reinterpret_cast<const bf16x8*>(xt + 2 * PPL + vb[sg]);
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    wh[cb] = *reinterpret_cast<const bf16x8*>(wst + (k * CB + cb) * 2048);
                    wl[cb] = *reinterpret_cast<const bf16x8*>(wst + (k * CB + cb) * 2048 + 1024);
                }
                if (++dx == a.kw) {
                    dx = 0;
                    if (++dy == a.kh) { dy = 0; ++ci; }
                }
                // D[filter][pixel]; no per-segment liveness guard (a dead segment's result is masked in the epilogue and a
                // guard would put every MFMA into its own basic block)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int sg = 0; sg < 2; ++sg) {
                        acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xh[sg], acc[cb][sg], 0, 0, 0);
                        KRK_CROSS(acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xl[sg], acc[cb][sg], 0, 0, 0);
                                  acc[cb][sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], xh[sg], acc[cb][sg], 0, 0, 0);)
                    }
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
        KRK_PH(a, 4);
    }

    // ------------------------------------------------------------------------------- epilogue (as conv_x3.hip, split outputs)
    auto store_tile = [&](auto actf) {
        // split NHWC (or split sequence rows): element index n*y_sn + row*y_sr + col*y_sc + filter
        __bf16* yh = reinterpret_cast<__bf16*>(a.y);
        __bf16* yl = yh + a.y_plane;
        const bool as_f32 = a.y_f32 != 0;
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
                        if (as_f32) {
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
#if defined(KRK_ABLATE) && !defined(KRK_BF16_ONE)
    KRK_PH(a, 5);
    KRK_PH_FLUSH(a, g_x3p_phases, 6);
#endif
}

template <int POOL, bool SB>
int launch_p(const X3Args& a, int cb, dim3 grid, size_t lds, hipStream_t s) {
#define KRK_LAUNCH(CB_)                                                                         \
    do {                                                                                        \
        auto kfn = conv_x3p_kernel<POOL, CB_, SB>;                                              \
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

}  // namespace

#if defined(KRK_ABLATE) && !defined(KRK_BF16_ONE)
int krk_phase_stats_x3p(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3p_phases), sizeof(g_x3p_phases)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_x3p_phases), z, sizeof(z)) != hipSuccess) return -1; }
    return 7;
}
#endif

#ifndef KRK_BF16_ONE
// tile copies per weight-stage boundary for this geometry, or 0 when the pipelined kernel does not cover it
int krk_conv_x3p_tps(int cchunk, int kb, int kb_last, int npix, int iw, int ntaps, int cout, size_t line_bytes) {
    if (cchunk != 16 || kb != 1 || kb_last != 1 || npix > 64 * NBMAX || iw >= 128 || line_bytes >= 0x7fffffffull) return 0;
    const int it = 4 / krk_x3_cb(cout);
    const int nstc = ntaps / it;                  // whole weight stages inside one chunk
    if (nstc < 4) return 0;
    const int nb = (npix + 63) / 64;
    const int tps = (nb + (nstc - 3) - 1) / (nstc - 3);
    return tps <= 4 ? tps : 0;
}
#endif

// same arguments as krk_launch_conv_x3 (split outputs only) + a.tps = krk_conv_x3p_tps(...) > 0
int KRK_FN(krk_launch_conv_x3p)(const X3Args& a, bool pool, hipStream_t s) {
    const int CBt = (a.Cout + 31) / 32;
    const int cb = krk_x3_cb(a.Cout);
    dim3 grid((unsigned)(a.tiles_w * a.tiles_h * a.N), (unsigned)((CBt + cb - 1) / cb));
    const int nb = (a.IH * a.IW + 63) / 64;
    // one (a.single_buf) or two tile buffers of (hi, lo) x 2 pieces + the weight ring
    const size_t lds = (size_t)(a.single_buf ? 4 : 8) * nb * 1024 + 3 * 8192;
    if (a.single_buf) return pool ? launch_p<1, true>(a, cb, grid, lds, s) : launch_p<0, true>(a, cb, grid, lds, s);
    return pool ? launch_p<1, false>(a, cb, grid, lds, s) : launch_p<0, false>(a, cb, grid, lds, s);
}
