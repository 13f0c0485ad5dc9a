// First convolution of a grayscale (Cin = 1) or colour (Cin = 3, round 5) line on the gfx950 bf16 matrix cores with split operands
// ("bf16x3", see conv_x3.hip).  Reference: kraken/lib/vgsl/layers.py ActConv2D.forward :842-860 applied to
// the (N, 1, H, W) input, with the following 2x2 MaxPool (:381-388) fused.
//
// With one input channel the GEMM's K axis is the kernel WINDOW, not channels: one MFMA K block of 16 is
// the 16 horizontal taps dx = 0..15 of one kernel row dy (taps >= kw carry zero weights), so a 3x13 kernel
// is 3 K blocks instead of the 39 K = 2 steps the fp32 path (conv_mfma.hip) issues.
//
//   MFMA      D[filter][pixel] += W[filter][dx] . X[dx][pixel],  X[dx][pixel] = in[row + dy][col + dx]
//   columns   the 32 MFMA columns of lane group c are pixels 4c + s (s = 0..3: four interleaved segments),
//             so the 8 taps a lane feeds are 8 CONSECUTIVE input pixels starting at 4c + 8*half + s: three
//             aligned ds_read_b64 give the 12-pixel window and v_alignbyte shifts produce the four segments
//   tile      4 waves x (2 output rows x 128 columns); an input row fetched once serves both output rows
//             (kernel rows dy and dy-1) and all four segments: 6 LDS reads per 24 MFMAs
//   weights   kh x (hi, lo) A fragments = kh*8 VGPRs, resident for the whole kernel (Cin = 1); with three input channels the
//             3 kh fragment pairs live in LDS (18 KB at kh = 3; 72 more VGPRs would halve the occupancy) and a lane re-reads
//             its 16 bytes per (channel, kernel row): 4 ds_read_b128 against 24 MFMAs.  The K axis then runs over
//             (channel, dy, dx): the tile holds the window of every channel, staged from the (N, 3, H, W) fp32 planes
//   staging   fp32 input -> (hi, lo) bf16 rows in LDS (6 KB per tile), register-prefetched one column
//             tile ahead; a workgroup walks all column tiles of its 8 output rows
//   epilogue  2x2 max-pool inside a lane (segments s, s+1 and the two rows), bias + activation, length
//             mask, split channels-last (NHWC) bf16 planes for conv_x3.hip
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LW = 144;     // LDS row: 128 output columns + 15 taps, padded to 8-byte reads
constexpr int TW = 128;     // output columns per tile
constexpr int TH = 8;       // output rows per tile (2 per wave)

// RELU: the activation is ReLU (every kraken recogniser): chosen at launch, so the epilogue is straight-line code
template <int KH, bool POOL, bool NHCW, bool RELU, int CIN>
__global__ void __launch_bounds__(256, 2) conv1_x3_kernel(const Conv1Args a) {
    constexpr int IH = TH + KH - 1;
    constexpr int NST = (IH * LW + 255) / 256;
    __shared__ __attribute__((aligned(16))) __bf16 tile[2][2][CIN][IH][LW];   // [buffer][plane][channel][row][column]
    __shared__ __attribute__((aligned(16))) float bias_s[32];
    // CIN > 1: the A fragments [channel][kernel row][plane][lane][8] (the order of a.wpack) in LDS
    // (dynamic: with three channels and five kernel rows the tile and these fragments are 72 KB together -- more than static LDS may be)
    extern __shared__ __attribute__((aligned(16))) __bf16 wlds[];        // [CIN * KH * 2 * 64 * 8] when CIN > 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, c = lane & 31;

    const int n = blockIdx.x / a.tiles_h;
    const int h0 = (blockIdx.x - n * a.tiles_h) * TH;
    const int len_in = a.len_in ? a.len_in[n] : a.W;
    const int len_out = a.len_out ? a.len_out[n] : a.Wy;
    const int wlim = POOL ? min(a.Wo, 2 * len_out) : min(a.Wo, len_out);

    // resident weights: A fragment of kernel row dy = 32 filters x 16 taps, lane (filter, half) holds taps 8*half..+7
    bf16x8 wh[KH], wl[KH];
    if constexpr (CIN == 1) {
#pragma unroll
        for (int dy = 0; dy < KH; ++dy) {
            wh[dy] = *reinterpret_cast<const bf16x8*>(a.wpack + ((size_t)(dy * 2 + 0) * 64 + lane) * 8);
            wl[dy] = *reinterpret_cast<const bf16x8*>(a.wpack + ((size_t)(dy * 2 + 1) * 64 + lane) * 8);
        }
    } else {
        for (int e = tid; e < CIN * KH * 2 * 64; e += 256)
            *reinterpret_cast<bf16x8*>(wlds + (size_t)e * 8) = *reinterpret_cast<const bf16x8*>(a.wpack + (size_t)e * 8);
    }
    // staging: element e = tid + 256*i of the IH x LW input window
    int s_off[NST], s_iw[NST];   // s_iw = column relative to w0, or a large negative number for a row outside the image
    const float* xin = a.x + (size_t)n * CIN * a.H * a.W;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i;
        const int ih = e / LW, iw = e - ih * LW;
        const int gh = h0 - a.ph + ih;
        const bool ok = e < IH * LW && gh >= 0 && gh < a.H;
        s_off[i] = gh * a.W + iw - a.pw;
        s_iw[i] = ok ? iw - a.pw : -(1 << 28);
    }
    // Vector memory of the tile loop is BRANCH-FREE (buffer instructions: an out-of-range offset reads 0 / drops the store):
    // with loads and stores under exec-mask branches the compiler cannot count what is in flight and waits with vmcnt(0) --
    // which, vmcnt retiring in order INCLUDING stores, holds the next tile until this tile's stores are acknowledged.
    constexpr unsigned kOOB = 0x7FFFF000u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, CIN * a.H * a.W * (int)sizeof(float), 0x00020000);
    const unsigned chan_b = (unsigned)(a.H * a.W) * 4u;      // bytes between the channel planes of a line
    float st[CIN][NST];
    auto gload = [&](int w0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int gw = w0 + s_iw[i];
            const bool ok = gw >= 0 && gw < len_in && !KRK_DBGBIT(a, 2);
#pragma unroll
            for (int ch = 0; ch < CIN; ++ch)
                st[ch][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok ? (unsigned)(s_off[i] + w0) * 4u + ch * chan_b : kOOB, 0, 0));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int ch = 0; ch < CIN; ++ch) {
            __bf16* t = &tile[buf][0][ch][0][0];
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = tid + 256 * i;
                if (e < IH * LW) {
                    const __bf16 h = (__bf16)st[ch][i];
                    t[e] = h;
                    t[CIN * IH * LW + e] = (__bf16)(st[ch][i] - (float)h);
                }
            }
        }
    };

    if (tid < 32) bias_s[tid] = a.bias[tid];
    // NHCW output: one descriptor per plane for THIS line's [Hy][Cout][pitch] block
    [[maybe_unused]] const int yline_b = a.Hy * a.Cout * a.y_pitch * (int)sizeof(__bf16);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t yrs_h = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.Hy * a.Cout * a.y_pitch, 0, yline_b, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t yrs_l = __builtin_amdgcn_make_buffer_rsrc(a.y + a.y_plane + (size_t)n * a.Hy * a.Cout * a.y_pitch, 0, yline_b, 0x00020000);
    gload(0);
    lstore(0);
    __syncthreads();

    for (int tw = 0; tw < a.tiles_w; ++tw) {
        const int w0 = tw * TW;
        const int buf = tw & 1;
        if (tw + 1 < a.tiles_w) gload(w0 + TW);

        f32x16 acc[2][4];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[o][s][r] = 0.f;

        if (w0 < wlim && !KRK_DBGBIT(a, 1)) {
#pragma unroll
          for (int ch = 0; ch < CIN; ++ch) {
#pragma unroll
            for (int i = 0; i < KH + 1; ++i) {
                if constexpr (CIN > 1) {
                    // this channel's fragments of kernel rows i and i - 1 (the two output rows of the wave), from LDS
                    if (i < KH) {
                        wh[i] = *reinterpret_cast<const bf16x8*>(wlds + ((size_t)((ch * KH + i) * 2 + 0) * 64 + lane) * 8);
                        wl[i] = *reinterpret_cast<const bf16x8*>(wlds + ((size_t)((ch * KH + i) * 2 + 1) * 64 + lane) * 8);
                    }
                }
                bf16x8 f[2][4];   // [plane][segment]
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const u32x2* src = reinterpret_cast<const u32x2*>(&tile[buf][p][ch][2 * wave + i][4 * c + 8 * half]);
                    const u32x2 q0 = src[0], q1 = src[1], q2 = src[2];
                    const unsigned d0 = q0[0], d1 = q0[1], d2 = q1[0], d3 = q1[1], d4 = q2[0], d5 = q2[1];
                    f[p][0] = __builtin_bit_cast(bf16x8, u32x4{d0, d1, d2, d3});
                    f[p][2] = __builtin_bit_cast(bf16x8, u32x4{d1, d2, d3, d4});
                    f[p][1] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbyte(d1, d0, 2), __builtin_amdgcn_alignbyte(d2, d1, 2),
                                                               __builtin_amdgcn_alignbyte(d3, d2, 2), __builtin_amdgcn_alignbyte(d4, d3, 2)});
                    f[p][3] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbyte(d2, d1, 2), __builtin_amdgcn_alignbyte(d3, d2, 2),
                                                               __builtin_amdgcn_alignbyte(d4, d3, 2), __builtin_amdgcn_alignbyte(d5, d4, 2)});
                }
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int dy = i - o;
                    if (dy < 0 || dy >= KH) continue;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[o][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[dy], f[0][s], acc[o][s], 0, 0, 0);
                        KRK_CROSS(acc[o][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[dy], f[1][s], acc[o][s], 0, 0, 0);
                                  acc[o][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[dy], f[0][s], acc[o][s], 0, 0, 0);)
                    }
                }
            }
          }
        }

        // ---- epilogue: lane = pixels w0 + 4c + s of rows h0 + 2*wave + o; register 4j+i = filter 8j + 4*half + i
        __bf16* yh = a.y;
        __bf16* yl = a.y + a.y_plane;
        // (re)read per tile: 16 VGPRs the K loop does not have to carry.  From LDS, not from memory: a vector-memory load here
        // would put `s_waitcnt vmcnt(small)` between the stores below, and vmcnt retires in order INCLUDING stores -- the wait for
        // a bias value then is a wait for the previous tile's stores to be acknowledged
        f32x4 bias4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias4[j] = *reinterpret_cast<const f32x4*>(bias_s + 8 * j + 4 * half);
        if constexpr (NHCW) {
            // "NHCW" planes [N][Hy][Cout][pitch]: a lane owns 2 (pooled) or 4 consecutive columns of each of its 16
            // filters -> one 4/8-byte store per filter, 128/256 bytes contiguous across the wave.  Columns between
            // the line's length and the pitch are written as zeros (the consumer stages whole 16-byte pieces).
            constexpr int NO = POOL ? 1 : 2;
            constexpr int NV = POOL ? 2 : 4;
            // the activation is chosen once per tile, not per element: ReLU (every kraken recogniser) is one v_max
            auto store_tile = [&](auto actf) {
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const int row = POOL ? (h0 >> 1) + wave : h0 + 2 * wave + o;
                    const int col0 = POOL ? (w0 >> 1) + 2 * c : w0 + 4 * c;
                    const int lim = min(len_out, a.Wy);
                    // buffer stores on per-line descriptors (hi and lo plane): 32-bit offsets inside one line's planes, a lane
                    // without an output (filter >= Cout, row / column outside) gets the out-of-range offset -- no branch, and
                    // no per-filter 64-bit address in vector registers across the K loop (hoisted there they cost 32 VGPRs
                    // and spilled: a scratch reload + vmcnt(0) in the middle of every tile's stores)
                    const unsigned lane_off = (unsigned)(((row * a.Cout + 4 * half) * a.y_pitch + col0) * (int)sizeof(__bf16));
                    const bool lane_ok = !(row >= a.Hy || col0 >= a.y_pitch || KRK_DBGBIT(a, 4));
                    const unsigned pitch_b = (unsigned)a.y_pitch * (unsigned)sizeof(__bf16);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f0 = (r & 3) + 8 * (r >> 2);             // this register's filter of lane half 0
                        __bf16 hv[NV], lv[NV];
#pragma unroll
                        for (int e = 0; e < NV; ++e) {
                            float v;
                            if (POOL) v = fmaxf(fmaxf(acc[0][2 * e][r], acc[0][2 * e + 1][r]), fmaxf(acc[1][2 * e][r], acc[1][2 * e + 1][r]));
                            else v = acc[o][e][r];
                            v = actf(v + bias4[r >> 2][r & 3]);
                            if (col0 + e >= lim) v = 0.f;
                            hv[e] = (__bf16)v;
                            lv[e] = (__bf16)(v - (float)hv[e]);
                        }
                        const unsigned vo = (lane_ok && f0 + 4 * half < a.Cout) ? lane_off + (unsigned)f0 * pitch_b : kOOB;
                        if (POOL) {
                            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bf16x2{hv[0], hv[1]}), yrs_h, vo, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bf16x2{lv[0], lv[1]}), yrs_l, vo, 0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, bf16x4{hv[0], hv[1], hv[2], hv[3]}), yrs_h, vo, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, bf16x4{lv[0], lv[1], lv[2], lv[3]}), yrs_l, vo, 0, 0);
                        }
                    }
                }
            };
            if constexpr (RELU) store_tile([](float v) { return fmaxf(v, 0.f); });
            else store_tile([&](float v) { return krk_act(v, a.act); });
        } else if constexpr (POOL) {
            const int prow = (h0 >> 1) + wave;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const int pcol = (w0 >> 1) + 2 * c + sp;
                const bool ok = prow < a.Hy && pcol < a.Wy;
                const size_t base = (size_t)n * a.y_sn + (size_t)prow * a.y_sr + (size_t)pcol * a.y_sc;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x4 hv, lv;
                    f32x4 fv;   // the same four values unsplit, for a GroupNorm consumer (y_f32)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * j + i;
                        float v = fmaxf(fmaxf(acc[0][2 * sp][r], acc[0][2 * sp + 1][r]), fmaxf(acc[1][2 * sp][r], acc[1][2 * sp + 1][r]));
                        v = krk_act(v + bias4[j][i], a.act);
                        if (pcol >= len_out) v = 0.f;
                        const __bf16 h = (__bf16)v;
                        hv[i] = h;
                        fv[i] = v;
                        lv[i] = (__bf16)(v - (float)h);
                    }
                    const int co = 8 * j + 4 * half;
                    if (ok && co < a.Cout) {
                        if (a.y_f32) {
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + base + co) = fv;
                        } else {
                            if (a.y_f32) {
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + base + co) = fv;
                            } else {
                                *reinterpret_cast<bf16x4*>(yh + base + co) = hv;
                                *reinterpret_cast<bf16x4*>(yl + base + co) = lv;
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int row = h0 + 2 * wave + o;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int col = w0 + 4 * c + s;
                    const bool ok = row < a.Ho && col < a.Wo;
                    const size_t base = (size_t)n * a.y_sn + (size_t)row * a.y_sr + (size_t)col * a.y_sc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf16x4 hv, lv;
                    f32x4 fv;   // the same four values unsplit, for a GroupNorm consumer (y_f32)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = krk_act(acc[o][s][4 * j + i] + bias4[j][i], a.act);
                            if (col >= len_out) v = 0.f;
                            const __bf16 h = (__bf16)v;
                            hv[i] = h;
                        fv[i] = v;
                            lv[i] = (__bf16)(v - (float)h);
                        }
                        const int co = 8 * j + 4 * half;
                        if (ok && co < a.Cout) {
                            if (a.y_f32) {
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + base + co) = fv;
                            } else {
                                *reinterpret_cast<bf16x4*>(yh + base + co) = hv;
                                *reinterpret_cast<bf16x4*>(yl + base + co) = lv;
                            }
                        }
                    }
                }
            }
        }

        if (tw + 1 < a.tiles_w) lstore(buf ^ 1);
        // LDS hand-over only.  __syncthreads() would put a full vmcnt(0) in front of the barrier, i.e. wait until this tile's
        // 32 stores per lane are acknowledged by memory before the next tile may start: the 503 MB this kernel writes then
        // drain with the matrix pipe idle (ablation: 0.127 ms without stores + 0.116 ms of stores = the 0.243 ms measured)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

template <int KH, int CIN>
int launch_kh(const Conv1Args& a, bool pool, hipStream_t s) {
    dim3 grid((unsigned)(a.N * a.tiles_h));
    const bool nhcw = a.y_pitch > 0;
    const bool relu = a.act == ACT_RELU;
    constexpr size_t dyn = CIN > 1 ? (size_t)CIN * KH * 2 * 64 * 8 * sizeof(__bf16) : 0;       // the weight fragments in LDS
#define KRK_C1(P_, N_, R_) do { \
        if (dyn) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_x3_kernel<KH, P_, N_, R_, CIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
        hipLaunchKernelGGL((conv1_x3_kernel<KH, P_, N_, R_, CIN>), grid, dim3(256), dyn, s, a); } while (0)
    if (pool && nhcw && relu) KRK_C1(true, true, true);
    else if (pool && nhcw) KRK_C1(true, true, false);
    else if (pool) KRK_C1(true, false, false);
    else if (nhcw && relu) KRK_C1(false, true, true);
    else if (nhcw) KRK_C1(false, true, false);
    else KRK_C1(false, false, false);
#undef KRK_C1
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

#ifndef KRK_BF16_ONE
bool krk_conv1_x3_supported(int Cin, int Cout, int kh, int kw, int sh, int sw, int dh, int dw) {
    // three channels (colour models): kernel rows 1, 3 and 5 (round 6: the fragments moved to dynamic LDS; 7 rows would need 97 KB: one workgroup per CU)
    // (a launch computes up to 32 filters; up to 64 are two launches on the two halves of the channels-last output: capi.hip)
    return (Cin == 1 || (Cin == 3 && kh <= 5)) && Cout <= 64 && Cout % 4 == 0 && (kh == 1 || kh == 3 || kh == 5 || (kh == 7 && Cin == 1)) && kw >= 1 &&
           kw <= 16 && sh == 1 && sw == 1 && dh == 1 && dw == 1;
}

#endif

int KRK_FN(krk_launch_conv1_x3)(const Conv1Args& a, bool pool, hipStream_t s) {
    if (a.N <= 0) return 0;
    if (a.Cin == 3) {
        switch (a.kh) {
            case 1: return launch_kh<1, 3>(a, pool, s);
            case 3: return launch_kh<3, 3>(a, pool, s);
            case 5: return launch_kh<5, 3>(a, pool, s);
            default: return -1;
        }
    }
    switch (a.kh) {
        case 1: return launch_kh<1, 1>(a, pool, s);
        case 3: return launch_kh<3, 1>(a, pool, s);
        case 5: return launch_kh<5, 1>(a, pool, s);
        case 7: return launch_kh<7, 1>(a, pool, s);       // (round 6: specs that open with Cr7,7,32 ran their first layer on the exact-f32 kernel)
        default: return -1;
    }
}
