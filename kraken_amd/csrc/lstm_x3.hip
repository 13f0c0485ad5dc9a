// Recurrent part of the (bi)directional LSTM with SPLIT-bf16 operands on the bf16 matrix cores
// (the "bf16x3" scheme of conv_x3.hip applied to  h_{t-1} . W_hh^T):  per step
//   gates = xproj[t] + h_hi.W_hi + h_lo.W_hi + h_hi.W_lo        (fp32 accumulate)
// Same reference semantics as lstm_rec.hip (kraken/lib/vgsl/layers.py:513-547); measured on BENCH-A the
// split recurrence adds < 2e-6 to the logit error of the fp32 recurrence over 150 steps.
//
// One workgroup = (16 lines, one direction), all time steps, no inter-workgroup traffic.
//   - h_{t-1} lives in LDS as split rows [line][k] (hi, lo): one ds_read_b128 per plane delivers the 8
//     consecutive K of a v_mfma_f32_16x16x32_bf16 A fragment (row stride padded: conflict-free);
//   - W_hh (hi, lo planes, same bytes as fp32) streams from L2 in B-fragment order, one K-block of 32
//     ahead, dwordx4 per lane;
//   - gate columns are interleaved (col = 4*unit + gate): the cell update happens inside DPP quads,
//     the cell state never leaves registers;
//   - h_t is written back to LDS as (hi, lo) and streamed to `out` as split planes [N][T][ndir*H] --
//     exactly what the next layer's split projection (conv_x3.hip) consumes: no fp32 round trip.
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {   // DPP quad_perm: 0xB1 = lane^1, 0x4E = lane^2
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

struct WPair {
    bf16x8 hi, lo;
};

template <int NBW, bool XPRE>
__device__ __forceinline__ void lstm_x3_loop(const LstmX3Args& a, unsigned char* hs, const int* lens_s, int Lmax,
                                             int wave, int lane, int dir, bool rev, int n0) {
    constexpr int M = 16;
    const int cl = lane & 15;
    const int gate = cl & 3, ul = cl >> 2;
    const int kq = lane >> 4;                 // which 8 of the 32 K of a block this lane holds
    const int RS = a.hrow;                    // bytes per h row (one line, one plane)
    const int plane = M * RS;                 // bytes per plane
    const int buf = 2 * plane;                // bytes per (hi, lo) buffer

    int irow[4], ilen[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        irow[r] = 4 * (lane >> 4) + r;
        ilen[r] = lens_s[irow[r]];
    }

    f32x4 acc[NBW];
    f32x4 xn[XPRE ? NBW : 1];
    float cst[NBW];          // cell state of (line 4*(lane>>4) + gate, unit): one per column block
#pragma unroll
    for (int j = 0; j < NBW; ++j) cst[j] = 0.f;

    // weights: [dir][kb][block][plane][lane][8] bf16
    const __bf16* wbase = a.wp + ((size_t)dir * a.NKB * a.NB * 2 * 64 + lane) * 8;
    const size_t kstride = (size_t)a.NB * 1024;
    const float gscale = (gate == 2) ? 2.f : 1.f;

    auto load_x = [&](int s, auto& dst) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool on = s < ilen[r];
            const int t = rev ? (ilen[r] - 1 - s) : s;
            const float* xr = a.xp + ((size_t)(n0 + irow[r]) * a.T + (on ? t : 0)) * a.xstride + (size_t)dir * a.G + cl;
#pragma unroll
            for (int j = 0; j < NBW; ++j) dst[j][r] = on ? xr[(size_t)(wave + 4 * j) * M] : 0.f;
        }
    };
    auto load_w = [&](int kb, WPair (&dst)[NBW]) {
        const __bf16* wk = wbase + (size_t)kb * kstride;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const __bf16* p = wk + (size_t)(wave + 4 * j) * 1024;
            dst[j].hi = *reinterpret_cast<const bf16x8*>(p);
            dst[j].lo = *reinterpret_cast<const bf16x8*>(p + 512);
        }
    };
    auto mma_block = [&](int kb, const unsigned char* hcur, const WPair (&w)[NBW]) {
        const unsigned char* ap = hcur + cl * RS + (kb * 32 + kq * 8) * 2;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + plane);
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w[j].hi, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w[j].hi, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w[j].lo, acc[j], 0, 0, 0);
    };

    WPair wa[NBW], wb[NBW];
    load_w(0, wa);
    if constexpr (XPRE) load_x(0, xn);
    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        if constexpr (XPRE) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[j] = xn[j];
            if (s + 1 < Lmax) load_x(s + 1, xn);
        } else {
            load_x(s, acc);
        }
        const unsigned char* hcur = hs + cur * buf;
        int kb = 0;
        for (; kb + 1 < a.NKB; kb += 2) {
            load_w(kb + 1, wb);
            mma_block(kb, hcur, wa);
            load_w(kb + 2 < a.NKB ? kb + 2 : 0, wa);      // wraps to block 0 of the next step
            mma_block(kb + 1, hcur, wb);
        }
        if (kb < a.NKB) {
            mma_block(kb, hcur, wa);
            load_w(0, wa);
        }

        // ---- gate non-linearities, cell update, h_t -> LDS as (hi, lo)
        // Lane q of a quad holds gate q of one unit for the tile's lines 4*(lane>>4) + r, r = 0..3.  After the
        // per-lane activation a 4x4 (register x lane) transpose inside the quad -- two DPP butterfly stages --
        // gives lane q all four gates of line r = q, so every lane updates ONE cell (no 4-fold redundancy).
        unsigned char* hnext = hs + (cur ^ 1) * buf;
        const int myrow = 4 * (lane >> 4) + gate;        // the line this lane owns after the transpose
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int unit = (wave + 4 * j) * 4 + ul;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gv = __builtin_amdgcn_rcpf(1.0f + __expf(-gscale * acc[j][r]));
                v[r] = (gate == 2) ? (2.f * gv - 1.f) : gv;
            }
            float b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {               // swap bit 0 of (register, lane)
                const float p = quad_perm<0xB1>(v[r ^ 1]);
                b[r] = (((r ^ gate) & 1) != 0) ? p : v[r];
            }
            float g4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {               // swap bit 1 of (register, lane)
                const float p = quad_perm<0x4E>(b[r ^ 2]);
                g4[r] = (((r ^ gate) & 2) != 0) ? p : b[r];
            }
            // g4[0..3] = sig(i), sig(f), tanh(g), sig(o) of (line myrow, unit)
            const float c = g4[1] * cst[j] + g4[0] * g4[2];
            cst[j] = c;
            const float h = g4[3] * krk_tanh(c);
            const __bf16 hh = (__bf16)h;
            __bf16* dst = reinterpret_cast<__bf16*>(hnext + myrow * RS) + unit;
            dst[0] = hh;
            *reinterpret_cast<__bf16*>(reinterpret_cast<unsigned char*>(dst) + plane) = (__bf16)(h - (float)hh);
        }
        __syncthreads();
        // ---- h_t -> split output planes out[plane][n][t][dir*H + k], coalesced 2-byte rows
        for (int i = wave; i < M; i += 4) {
            const int li = lens_s[i];
            if (s < li) {
                const int t = rev ? (li - 1 - s) : s;
                const size_t o = ((size_t)(n0 + i) * a.T + t) * a.ostride + (size_t)dir * a.H;
                const __bf16* src = reinterpret_cast<const __bf16*>(hnext + i * RS);
                for (int k = lane; k < a.H; k += 64) {
                    a.out[o + k] = src[k];
                    a.out[a.out_plane + o + k] = *reinterpret_cast<const __bf16*>(reinterpret_cast<const unsigned char*>(src + k) + plane);
                }
            }
        }
        cur ^= 1;
    }
}

__device__ __forceinline__ void lstm_x3_idle(const LstmX3Args& a, const unsigned char* hs, const int* lens_s, int Lmax,
                                             int wave, int lane, int dir, bool rev, int n0) {
    constexpr int M = 16;
    const int RS = a.hrow, plane = M * RS, buf = 2 * plane;
    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        const unsigned char* hnext = hs + (cur ^ 1) * buf;
        __syncthreads();
        for (int i = wave; i < M; i += 4) {
            const int li = lens_s[i];
            if (s < li) {
                const int t = rev ? (li - 1 - s) : s;
                const size_t o = ((size_t)(n0 + i) * a.T + t) * a.ostride + (size_t)dir * a.H;
                const __bf16* src = reinterpret_cast<const __bf16*>(hnext + i * RS);
                for (int k = lane; k < a.H; k += 64) {
                    a.out[o + k] = src[k];
                    a.out[a.out_plane + o + k] = *reinterpret_cast<const __bf16*>(reinterpret_cast<const unsigned char*>(src + k) + plane);
                }
            }
        }
        cur ^= 1;
    }
}

template <int MAXB, bool XPRE>
__global__ void __launch_bounds__(256, 1) lstm_x3_kernel(const LstmX3Args a) {
    constexpr int M = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    unsigned char* hs = smem8;                                       // [2 buffers][2 planes][16][hrow]
    int* lens_s = reinterpret_cast<int*>(smem8 + 4 * M * a.hrow);   // [16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const bool rev = (a.dirmode == 1) || (a.dirmode == 2 && dir == 1);
    const int n0 = blockIdx.x * M;

    if (tid < M) {
        const int n = n0 + tid;
        int l = 0;
        if (n < a.N) l = a.lens ? min(max(a.lens[n], 0), a.T) : a.T;
        lens_s[tid] = l;
    }
    for (int e = tid; e < M * a.hrow; e += 256) reinterpret_cast<unsigned int*>(hs)[e] = 0u;   // 4*M*hrow bytes
    __syncthreads();
    int Lmax = 0;
    for (int i = 0; i < M; ++i) Lmax = max(Lmax, lens_s[i]);

    const int nb_mine = (a.NB - wave + 3) / 4;
    if (nb_mine == MAXB) {
        lstm_x3_loop<MAXB, XPRE>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
    } else {
        if constexpr (MAXB > 1) lstm_x3_loop<MAXB - 1, XPRE>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
        else lstm_x3_idle(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
    }
}

template <int MAXB, bool XPRE>
int launch_one(const LstmX3Args& a, hipStream_t s) {
    dim3 grid((unsigned)((a.N + 15) / 16), (unsigned)a.ndir);
    const size_t lds = (size_t)4 * 16 * a.hrow + 16 * sizeof(int);
    auto kfn = lstm_x3_kernel<MAXB, XPRE>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// a.NB = 4*Hp/16 column blocks of 16 gate columns; a.NKB = K-blocks of 32 (Hp padded to a multiple of 32).
int krk_launch_lstm_x3(const LstmX3Args& a, hipStream_t s) {
    const int per_wave = (a.NB + 3) / 4;
#define KRK_CASE(B_) case B_: return launch_one<B_, true>(a, s)
    switch (per_wave) {
        KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7); KRK_CASE(8);
        KRK_CASE(9); KRK_CASE(10); KRK_CASE(11); KRK_CASE(12); KRK_CASE(13); KRK_CASE(14); KRK_CASE(15); KRK_CASE(16);
        default: return -4;
    }
#undef KRK_CASE
}
