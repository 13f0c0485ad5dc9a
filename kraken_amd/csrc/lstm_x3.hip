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

struct WPair {
    bf16x8 hi, lo;
};

// h_t (hi, lo rows in LDS) -> split output planes out[plane][n][t][dir*H + k]; 16-byte pieces when H % 8 == 0
template <int NW>
__device__ __forceinline__ void lstm_x3_store(const LstmX3Args& a, const unsigned char* hnext, const int* lens_s, int s,
                                              int wave, int lane, int dir, bool rev, int n0) {
    constexpr int M = 16;
    constexpr int LPW = M / NW;                          // lines each wave copies out
    const int RS = a.hrow, plane = M * RS;
    if ((a.H & 7) == 0) {
        const int per_line = a.H >> 3;                   // 16-byte pieces per line per plane
        const int total = LPW * per_line * 2;            // this wave: LPW lines x 2 planes
        for (int e = lane; e < total; e += 64) {
            const int pl = e / (LPW * per_line), r = e - pl * LPW * per_line;
            const int li = r / per_line, q = r - li * per_line;
            const int i = wave * LPW + li;
            const int len = lens_s[i];
            if (s < len) {
                const int t = rev ? (len - 1 - s) : s;
                // K-blocked sequence rows [feature/8][line*T + t][8] (what gemm_x3.hip streams)
                const size_t o = ((size_t)(dir * per_line + q) * ((size_t)a.N * a.T) + (size_t)(n0 + i) * a.T + t) * 8;
                const f32x4 v = *reinterpret_cast<const f32x4*>(hnext + pl * plane + i * RS + q * 16);
                *reinterpret_cast<f32x4*>(a.out + pl * a.out_plane + o) = v;
            }
        }
    } else {
        for (int i = wave * LPW; i < wave * LPW + LPW; ++i) {
            const int len = lens_s[i];
            if (s < len) {
                const int t = rev ? (len - 1 - s) : s;
                const size_t rowi = (size_t)(n0 + i) * a.T + t, rows = (size_t)a.N * a.T;
                const __bf16* src = reinterpret_cast<const __bf16*>(hnext + i * RS);
                for (int k = lane; k < a.H; k += 64) {
                    const int f = dir * a.H + k;
                    const size_t o = ((size_t)(f >> 3) * rows + rowi) * 8 + (f & 7);
                    a.out[o] = src[k];
                    a.out[a.out_plane + o] = *reinterpret_cast<const __bf16*>(reinterpret_cast<const unsigned char*>(src + k) + plane);
                }
            }
        }
    }
}

// Orientation: D = W . h^T, i.e. the MFMA's A operand is the weight fragment (16 gate columns x 32 K) and B is
// h (32 K x 16 lines).  With gate columns interleaved (col = 4*unit + gate) the D fragment of lane l holds
// rows 4*(l>>4) + r = the FOUR GATES (r = i,f,g,o) of unit (l>>4) of the block, for line l&15: the cell update
// is purely per-lane (no cross-lane traffic), and xproj[t] for a block is ONE 16-byte load per lane.
template <int NW, int NBW, bool XPRE>
__device__ __forceinline__ void lstm_x3_loop(const LstmX3Args& a, unsigned char* hs, const int* lens_s, int Lmax,
                                             int wave, int lane, int dir, bool rev, int n0) {
    constexpr int M = 16;
    const int line = lane & 15;               // the line (B/D column) this lane owns
    const int us = lane >> 4;                 // unit inside a block (D rows 4*us..4*us+3), also the K octet of operands
    const int RS = a.hrow;                    // bytes per h row (one line, one plane)
    const int plane = M * RS;                 // bytes per plane
    const int buf = 2 * plane;                // bytes per (hi, lo) buffer
    const int mylen = lens_s[line];

    f32x4 acc[NBW];
    float cst[NBW];                           // cell state of (line, unit): one per column block
#pragma unroll
    for (int j = 0; j < NBW; ++j) cst[j] = 0.f;

    // weights: [dir][kb][block][plane][lane][8] bf16; lane l: gate column l&15 of the block, K octet l>>4
    const __bf16* wbase = a.wp + ((size_t)dir * a.NKB * a.NB * 2 * 64 + lane) * 8;
    const size_t kstride = (size_t)a.NB * 1024;
    const float* xrow = a.xp + (size_t)(n0 + line) * a.T * a.xstride + (size_t)dir * a.G + us * 4;

    auto load_x = [&](int s, auto& dst) {
        const bool on = s < mylen;
        const int t = on ? (rev ? (mylen - 1 - s) : s) : 0;
        const float* xr = xrow + (size_t)t * a.xstride;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on) v = *reinterpret_cast<const f32x4*>(xr + (wave + NW * j) * M);
            dst[j] = v;
        }
    };
    auto load_w = [&](int kb, WPair (&dst)[NBW]) {
        if (a.dbg & 1) return;
        const __bf16* wk = wbase + (size_t)kb * kstride;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const __bf16* p = wk + (size_t)(wave + NW * j) * 1024;
            dst[j].hi = *reinterpret_cast<const bf16x8*>(p);
            dst[j].lo = *reinterpret_cast<const bf16x8*>(p + 512);
        }
    };
    auto mma_block = [&](int kb, const unsigned char* hcur, const WPair (&w)[NBW]) {
        if (a.dbg & 4) return;
        const unsigned char* hp = hcur + line * RS + (kb * 32 + us * 8) * 2;
        const bf16x8 hh = *reinterpret_cast<const bf16x8*>(hp);
        const bf16x8 hl = *reinterpret_cast<const bf16x8*>(hp + plane);
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j].hi, hh, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j].hi, hl, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j].lo, hh, acc[j], 0, 0, 0);
    };

    WPair wa[NBW], wb[NBW];
    if (a.dbg & 1) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) { wa[j].hi = bf16x8{}; wa[j].lo = bf16x8{}; wb[j].hi = bf16x8{}; wb[j].lo = bf16x8{}; }
    }
    int cur = 0;
    // one time step; `xbuf` holds xproj of this step and is refilled with the next step's after the last weight
    // load (vmcnt retires in order: an HBM-latency load in front of weight loads would stall their MFMAs)
    auto step = [&](int s, f32x4 (&xbuf)[XPRE ? NBW : 1]) {
        if constexpr (XPRE) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[j] = xbuf[j];
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
        if constexpr (XPRE) {
            if (s + 1 < Lmax && !(a.dbg & 8)) load_x(s + 1, xbuf);
        }

        // ---- gate non-linearities, cell update (per lane: one (line, unit)), h_t -> LDS as (hi, lo)
        unsigned char* hnext = hs + (cur ^ 1) * buf;
        if (!(a.dbg & 2))
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int unit = (wave + NW * j) * 4 + us;
            const float gi = krk_sigmoid(acc[j][0]);
            const float gf = krk_sigmoid(acc[j][1]);
            const float gg = krk_tanh(acc[j][2]);
            const float go = krk_sigmoid(acc[j][3]);
            const float c = gf * cst[j] + gi * gg;
            cst[j] = c;
            const float h = go * krk_tanh(c);
            const __bf16 hh = (__bf16)h;
            __bf16* dst = reinterpret_cast<__bf16*>(hnext + line * RS) + unit;
            dst[0] = hh;
            *reinterpret_cast<__bf16*>(reinterpret_cast<unsigned char*>(dst) + plane) = (__bf16)(h - (float)hh);
        }
        __syncthreads();
        lstm_x3_store<NW>(a, hnext, lens_s, s, wave, lane, dir, rev, n0);
        cur ^= 1;
    };

    f32x4 xa[XPRE ? NBW : 1];
    load_w(0, wa);
    if constexpr (XPRE) load_x(0, xa);
    for (int s = 0; s < Lmax; ++s) step(s, xa);
}

template <int NW>
__device__ __forceinline__ void lstm_x3_idle(const LstmX3Args& a, const unsigned char* hs, const int* lens_s, int Lmax,
                                             int wave, int lane, int dir, bool rev, int n0) {
    constexpr int M = 16;
    const int buf = 2 * M * a.hrow;
    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        __syncthreads();
        lstm_x3_store<NW>(a, hs + (cur ^ 1) * buf, lens_s, s, wave, lane, dir, rev, n0);
        cur ^= 1;
    }
}

template <int NW, int MAXB, bool XPRE>
__global__ void __launch_bounds__(64 * NW, 1) lstm_x3_kernel(const LstmX3Args a) {
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
    for (int e = tid; e < M * a.hrow; e += 64 * NW) reinterpret_cast<unsigned int*>(hs)[e] = 0u;   // 4*M*hrow bytes
    __syncthreads();
    int Lmax = 0;
    for (int i = 0; i < M; ++i) Lmax = max(Lmax, lens_s[i]);

    const int nb_mine = (a.NB - wave + NW - 1) / NW;
    if (nb_mine == MAXB) {
        lstm_x3_loop<NW, MAXB, XPRE>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
    } else {
        if constexpr (MAXB > 1) lstm_x3_loop<NW, MAXB - 1, XPRE>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
        else lstm_x3_idle<NW>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
    }
}

template <int NW, int MAXB, bool XPRE>
int launch_one(const LstmX3Args& a, hipStream_t s) {
    dim3 grid((unsigned)((a.N + 15) / 16), (unsigned)a.ndir);
    const size_t lds = (size_t)4 * 16 * a.hrow + 16 * sizeof(int);
    auto kfn = lstm_x3_kernel<NW, MAXB, XPRE>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(64 * NW), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// a.NB = 4*Hp/16 column blocks of 16 gate columns; a.NKB = K-blocks of 32 (Hp padded to a multiple of 32).
// Waves per workgroup: 8 (two per SIMD, <= 256 VGPRs each) once a wave would own more than 4 blocks -- the
// second wave hides the L2 weight stream and the serial gate math of the first; 4 otherwise.
int krk_launch_lstm_x3(const LstmX3Args& a, hipStream_t s) {
    int nw = a.NB > 16 && a.NB <= 64 ? 8 : 4;
    if (const char* e = getenv("KRK_LSTM_NW")) nw = atoi(e) == 8 && a.NB <= 64 ? 8 : 4;
    if (nw == 8) {
        const int per_wave = (a.NB + 7) / 8;
#define KRK_CASE(B_) case B_: return launch_one<8, B_, true>(a, s)
        switch (per_wave) {
            KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7); KRK_CASE(8);
            default: return -4;
        }
#undef KRK_CASE
    }
    const int per_wave = (a.NB + 3) / 4;
#define KRK_CASE(B_) case B_: return launch_one<4, B_, true>(a, s)
    switch (per_wave) {
        KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7); KRK_CASE(8);
        KRK_CASE(9); KRK_CASE(10); KRK_CASE(11); KRK_CASE(12); KRK_CASE(13); KRK_CASE(14); KRK_CASE(15); KRK_CASE(16);
        default: return -4;
    }
#undef KRK_CASE
}
