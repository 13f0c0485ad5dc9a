// Implicit-GEMM convolution / projection on the gfx950 f32 matrix cores.
//
// Replaces (reference, kraken/lib/vgsl/layers.py): ActConv2D.forward :842-860
// (torch.nn.Conv2d + bias + activation), the directly following MaxPool 2x2/2
// (:381-388, fused), the height->channel Reshape (:313-335, fused as the
// channels-last epilogue), the input projection of nn.LSTM (:507-511) and the
// Linear of LinSoftmax (:708-722).
//
// Work decomposition (one 256-thread workgroup = 4 waves):
//   tile      = (8/SR) output rows x (32*SR) output columns of ONE line, CBW blocks of 32 filters
//   wave      = 2 segments of 32 consecutive output pixels (for the fused pool: the two rows of a
//               pooling pair), all CBW filter blocks -> 2*CBW accumulators of 32x32 f32
//   K axis    = (channel, dy, dx) of one LDS-resident channel chunk, two K per v_mfma_f32_32x32x2_f32;
//               the LDS offset of each K is looked up in a table so that kernel size, stride and
//               dilation are data, not code
//   operands  = input pixels from the LDS tile (lanes 0-31 read 32 consecutive floats: conflict-free),
//               weights straight from global memory in fragment order, four K-steps per lane
//               contiguous (one dwordx4 = 1 KB per wave instruction; the weights stay L2-resident)
//   pipeline  = K-steps are processed in groups of four; offsets, pixels and weights of group g+1
//               (weights: g+2) are fetched while the MFMAs of group g issue
//   staging   = all global loads of a chunk are issued before the first LDS write (no load->store
//               serialisation); sequence inputs use 16-byte loads along the feature axis
// Masked-padding semantics: input columns >= len_in[n] read as zero, stored columns >= len_out[n]
// are written as zero (see include/kraken_amd.h).
#include "common.h"

namespace {


// e / d for 0 <= e < 2^23, d > 0 via one float multiply and a fix-up (no integer division)
__device__ __forceinline__ int fast_div(int e, int d, float inv, int& rem) {
    int q = (int)((float)e * inv);
    int r = e - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

template <int IN_SEQ, int OUT_SEQ, int POOL, int CBW>
__global__ void __launch_bounds__(256, 2) conv_f32_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* otab = reinterpret_cast<int*>(smem);   // [2][KSG4]: offsets of K = 2*ks + half, ks-major per half
    float* tile = smem + a.otab_floats;

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
    const int cb0 = blockIdx.y * CBW;

    const int len_in = a.len_in ? a.len_in[n] : a.W;
    const int len_out = a.len_out ? a.len_out[n] : a.Wy;
    // conv-output columns that still carry data after masking
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

    // K -> LDS offset table (identical for every channel chunk); KS4 = padded K-steps (+2 spare groups)
    const int KS4 = a.KS4;
    for (int k = tid; k < 2 * KS4; k += 256) {
        const int hh = k / KS4, ks = k - hh * KS4;
        const int kidx = 2 * ks + hh;
        int off = 0;
        if (kidx < a.Kc) {
            const int kk = a.kh * a.kw;
            const int c = kidx / kk, rem = kidx - c * kk;
            const int dy = rem / a.kw, dx = rem - dy * a.kw;
            off = c * a.PS + dy * a.dh * a.RS + dx * a.dw;
        }
        otab[k] = off;
    }

    f32x16 acc[CBW][2];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][s][r] = 0.f;

    int boff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) boff[s] = srow[s] * a.sh * a.RS + (scol[s] + px) * a.sw;

    const int* otab_h = otab + half * KS4;

    for (int ci = 0; ci < a.nchunks; ++ci) {
        // ------------------------------------------------ stage chunk ci: loads first, then LDS writes
        if (IN_SEQ) {
            // rows of [W][Cin]: 16 lanes x 16 B cover 64 consecutive features of one pixel
            const int cc = a.cchunk;
            const int c4 = (lane & 15) * 4;
            const int pl = wave * 4 + (lane >> 4);          // pixel within a group of 16
            const int gc = ci * cc + c4;
            const bool cok = c4 < cc;
            if (a.vec4) {
                f32x4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gp = w0 + i * 16 + pl;
                    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (cok && i * 16 < TW && gp < a.W && gc < a.Cin)     // Cin % 4 == 0: whole quad valid
                        v[i] = *reinterpret_cast<const f32x4*>(a.x + (size_t)gp * a.Cin + gc);
                }
                __syncthreads();   // previous chunk fully consumed
                if (cok) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int p = i * 16 + pl;
#pragma unroll
                        for (int j = 0; j < 4; ++j) tile[(c4 + j) * a.PS + p] = v[i][j];
                    }
                }
            } else {
                __syncthreads();
                const int total = TW * cc;
                const float inv = 1.0f / (float)cc;
                for (int e = tid; e < total; e += 256) {
                    int c;
                    const int p = fast_div(e, cc, inv, c);
                    const int gp = w0 + p, g2 = ci * cc + c;
                    float v = 0.f;
                    if (gp < a.W && g2 < a.Cin) v = a.x[(size_t)gp * a.Cin + g2];
                    tile[c * a.PS + p] = v;
                }
            }
        } else {
            const int gh0 = h0 * a.sh - a.ph, gw0 = w0 * a.sw - a.pw;
            const int per_c = a.IH * a.IW;
            const int total = a.cchunk * per_c;
            const float inv_pc = 1.0f / (float)per_c, inv_iw = 1.0f / (float)a.IW;
            const float* xn = a.x + (size_t)n * a.Cin * a.H * a.W;
            __syncthreads();   // previous chunk fully consumed
            // batches of SB loads in flight per thread, then their LDS writes
            constexpr int SB = 16;
            for (int i0 = 0; i0 * 256 < total; i0 += SB) {
                float v[SB];
                int dst[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int e = tid + 256 * (i0 + i);
                    v[i] = 0.f;
                    dst[i] = -1;
                    if (e < total) {
                        int r, iw;
                        const int c = fast_div(e, per_c, inv_pc, r);
                        const int ih = fast_div(r, a.IW, inv_iw, iw);
                        const int gc = ci * a.cchunk + c, gh = gh0 + ih, gw = gw0 + iw;
                        dst[i] = c * a.PS + ih * a.RS + iw;
                        if (gc < a.Cin && gh >= 0 && gh < a.H && gw >= 0 && gw < len_in)
                            v[i] = xn[((size_t)gc * a.H + gh) * a.W + gw];
                    }
                }
#pragma unroll
                for (int i = 0; i < SB; ++i)
                    if (dst[i] >= 0) tile[dst[i]] = v[i];
            }
        }
        __syncthreads();
        if (!any_live) continue;

        // ------------------------------------------------ MFMA loop, groups of 4 K-steps, pipelined
        const int ngroups = (ci + 1 == a.nchunks) ? a.KSG_last : a.KSG;
        // weights: [chunk][group][cb][lane][4]
        const float* wp = a.wpack + (((size_t)ci * a.KSGpad) * a.CBpad + cb0) * 256 + lane * 4;
        const size_t wstep = (size_t)a.CBpad * 256;

        auto load_w = [&](int g, f32x4 (&w)[CBW]) {
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) w[cb] = *reinterpret_cast<const f32x4*>(wp + (size_t)g * wstep + cb * 256);
        };
        auto load_off = [&](int g) { return *reinterpret_cast<const int4*>(otab_h + 4 * g); };
        auto load_b = [&](const int4& o, float (&b)[2][4]) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                b[s][0] = tile[o.x + boff[s]];
                b[s][1] = tile[o.y + boff[s]];
                b[s][2] = tile[o.z + boff[s]];
                b[s][3] = tile[o.w + boff[s]];
            }
        };
        auto mma4 = [&](const float (&b)[2][4], const f32x4 (&w)[CBW]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (live[s]) {
                            if (OUT_SEQ)  // D[pixel][filter]
                                acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s][e], w[cb][e], acc[cb][s], 0, 0, 0);
                            else          // D[filter][pixel]
                                acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cb][e], b[s][e], acc[cb][s], 0, 0, 0);
                        }
        };

        // the offset table and the weight pack carry two zero groups of slack: no tail branches
        f32x4 wa[CBW], wb[CBW];
        float ba[2][4], bb[2][4];
        load_w(0, wa);
        load_w(1, wb);
        int4 o_next = load_off(1);
        {
            const int4 o0 = load_off(0);
            load_b(o0, ba);
        }
        for (int g = 0; g < ngroups; g += 2) {
            load_b(o_next, bb);                 // pixels of group g+1
            o_next = load_off(g + 2);
            mma4(ba, wa);
            load_w(g + 2, wa);
            load_b(o_next, ba);                 // pixels of group g+2
            o_next = load_off(g + 3);
            if (g + 1 < ngroups) mma4(bb, wb);
            load_w(g + 3, wb);
        }
    }

    // ------------------------------------------------------------- epilogues
    if (!OUT_SEQ && a.y_split) {
        // split-bf16 channels-last output for the bf16x3 kernels: element n*y_sn + row*y_sr + col*y_sc + filter
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        __bf16* yh = reinterpret_cast<__bf16*>(a.y_split);
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
            for (int cb = 0; cb < CBW; ++cb) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int co = (cb0 + cb) * 32 + 8 * rq + 4 * half;
                    bf16x4 hv, lv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[cb][s][4 * rq + i];
                        if (POOL) {
                            v = fmaxf(v, acc[cb][1][4 * rq + i]);
                            v = fmaxf(v, __shfl_xor(v, 1));
                        }
                        v = krk_act(v + a.bias[min(co + i, a.CBpad * 32 - 1)], a.act);
                        if (col >= len_out) v = 0.f;
                        const __bf16 h = (__bf16)v;
                        hv[i] = h;
                        lv[i] = (__bf16)(v - (float)h);
                    }
                    if (st && co < a.Cout) {
                        if (a.y_cs == 1) {
                            *reinterpret_cast<bf16x4*>(yh + base + co) = hv;
                            *reinterpret_cast<bf16x4*>(yl + base + co) = lv;
                        } else {        // [N][H][C][pitch] planes for conv_taps_x3.hip: consecutive lanes are consecutive columns
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (co + i < a.Cout) {
                                    yh[base + (size_t)(co + i) * a.y_cs] = hv[i];
                                    yl[base + (size_t)(co + i) * a.y_cs] = lv[i];
                                }
                        }
                    }
                }
            }
        }
        return;
    }
    if (POOL) {
        // rows srow[0], srow[1] form one pooling pair; adjacent lanes form the column pair
        if (!inb[0]) return;
        const int pr = (h0 + srow[0]) >> 1;
        const int pc = (w0 + scol[0] + px) >> 1;
        const bool st = !(px & 1) && pr < a.Hy && pc < a.Wy;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (cb0 + cb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = fmaxf(acc[cb][0][r], acc[cb][1][r]);
                v = fmaxf(v, __shfl_xor(v, 1));
                if (st && co < a.Cout) {
                    v = krk_act(v + a.bias[co], a.act);  // monotone activations commute with max
                    if (pc >= len_out) v = 0.f;
                    a.y[(((size_t)n * a.Cout + co) * a.Hy + pr) * a.Wy + pc] = v;
                }
            }
        }
    } else if (OUT_SEQ) {
        // channels-last: y[n][col][row*Cout + filter]  (row*Cout+filter == the reference's h*C+c)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!inb[s]) continue;
            const int row = h0 + srow[s];
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
                const int co = (cb0 + cb) * 32 + px;
                if (co >= a.Cout) continue;
                const float bv = a.bias[co];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = w0 + scol[s] + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (col < a.Wo) {
                        float v = krk_act(acc[cb][s][r] + bv, a.act);
                        if (col >= len_out) v = 0.f;
                        a.y[(((size_t)n * a.Wo + col) * a.Ho + row) * a.Cout + co] = v;
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!inb[s]) continue;
            const int row = h0 + srow[s];
            const int col = w0 + scol[s] + px;
            if (col >= a.Wo) continue;
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = (cb0 + cb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < a.Cout) {
                        float v = krk_act(acc[cb][s][r] + a.bias[co], a.act);
                        if (col >= len_out) v = 0.f;
                        a.y[(((size_t)n * a.Cout + co) * a.Ho + row) * a.Wo + col] = v;
                    }
                }
            }
        }
    }
}

template <int IN_SEQ, int OUT_SEQ, int POOL>
int launch_cbw(const ConvArgs& a, int cbw, dim3 grid, size_t lds, hipStream_t s) {
#define KRK_LAUNCH(CBW_)                                                                        \
    do {                                                                                        \
        auto kfn = conv_f32_kernel<IN_SEQ, OUT_SEQ, POOL, CBW_>;                                \
        if (lds > 48 * 1024)                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);                                    \
    } while (0)
    switch (cbw) {
        case 1: KRK_LAUNCH(1); break;
        case 2: KRK_LAUNCH(2); break;
        default: KRK_LAUNCH(4); break;
    }
#undef KRK_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

int krk_launch_conv(const ConvArgs& a, bool in_seq, bool out_seq, bool pool, hipStream_t s) {
    const int CB = (a.Cout + 31) / 32;
    const int cbw = CB >= 4 ? 4 : (CB >= 2 ? 2 : 1);
    dim3 grid((unsigned)(a.tiles_w * a.tiles_h * a.N), (unsigned)((CB + cbw - 1) / cbw));
    const size_t lds = ((size_t)a.otab_floats + (size_t)a.cchunk * a.PS) * sizeof(float);
    if (in_seq) {
        if (pool) return -1;
        return out_seq ? launch_cbw<1, 1, 0>(a, cbw, grid, lds, s) : launch_cbw<1, 0, 0>(a, cbw, grid, lds, s);
    }
    if (pool) return out_seq ? -1 : launch_cbw<0, 0, 1>(a, cbw, grid, lds, s);
    return out_seq ? launch_cbw<0, 1, 0>(a, cbw, grid, lds, s) : launch_cbw<0, 0, 0>(a, cbw, grid, lds, s);
}
