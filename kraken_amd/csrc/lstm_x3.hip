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
template <int NW, int M>
__device__ __forceinline__ void lstm_x3_store(const LstmX3Args& a, const unsigned char* hnext, const int* lens_s, int s,
                                              int wave, int lane, int dir, bool rev, int n0) {
    constexpr int LPW = M / NW;                          // lines each wave copies out
    const int RS = a.hrow, plane = M * RS;
    // row of (line n, step t): line-major, or tile-time-major (16-line tiles) for a gemm_x3 consumer that keeps that order
    const size_t rows = a.otiled ? (size_t)((a.N + 15) / 16 * 16) * a.T : (size_t)a.N * a.T;
    auto row_of = [&](int n, int t) -> size_t {
        return a.otiled ? ((size_t)(n >> 4) * a.T + t) * 16 + (n & 15) : (size_t)n * a.T + t;
    };
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
                const size_t o = ((size_t)(dir * per_line + q) * rows + row_of(n0 + i, t)) * 8;
                const f32x4 v = *reinterpret_cast<const f32x4*>(hnext + pl * plane + i * RS + q * 16);
                *reinterpret_cast<f32x4*>(a.out + pl * a.out_plane + o) = v;
            }
        }
    } else {
        for (int i = wave * LPW; i < wave * LPW + LPW; ++i) {
            const int len = lens_s[i];
            if (s < len) {
                const int t = rev ? (len - 1 - s) : s;
                const size_t rowi = row_of(n0 + i, t);
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
template <int NW, int G, int NBW, bool XPRE>
__device__ __forceinline__ void lstm_x3_loop(const LstmX3Args& a, unsigned char* hs, const int* lens_s, int Lmax,
                                             int wave, int lane, int dir, bool rev, int n0) {
    constexpr int M = 16 * G;                 // lines per workgroup: G groups of 16 share every weight fragment
    const int line = lane & 15;               // the line (B/D column) this lane owns inside each group
    const int us = lane >> 4;                 // unit inside a block (D rows 4*us..4*us+3), also the K octet of operands
    const int RS = a.hrow;                    // bytes per h row (one line, one plane)
    const int plane = M * RS;                 // bytes per plane
    const int buf = 2 * plane;                // bytes per (hi, lo) buffer
    int mylen[G];
#pragma unroll
    for (int g = 0; g < G; ++g) mylen[g] = lens_s[16 * g + line];

    f32x4 acc[G][NBW];
    float cst[G][NBW];                        // cell state of (line, unit): one per column block
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < NBW; ++j) cst[g][j] = 0.f;

    // weights: [dir][kb][block][plane][lane][8] bf16; lane l: gate column l&15 of the block, K octet l>>4
    const __bf16* wbase = a.wp + ((size_t)dir * a.NKB * a.NB * 2 * 64 + lane) * 8;
    const size_t kstride = (size_t)a.NB * 1024;
    // plain rows: (line n, step t) at n*T + t; tile-time-major rows (gemm_x3.hip): ((n/16)*T + t)*16 + n%16
    const float* xrow0 = a.xp + (a.xtiled ? ((size_t)n0 * a.T + line) : (size_t)(n0 + line) * a.T) * a.xstride + (size_t)dir * a.G + us * 4;
    const size_t xg_step = a.xtiled ? 16 : 1, xg_tile = (size_t)16 * a.T;   // rows per time step / per 16-line group

    auto load_x = [&](int s, int g, f32x4 (&dst)[NBW]) {
        const bool on = s < mylen[g];
        const int t = on ? (rev ? (mylen[g] - 1 - s) : s) : 0;
        const float* xr = xrow0 + ((size_t)g * xg_tile + (size_t)t * xg_step) * a.xstride;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + (wave + NW * j) * 16));   // read once: keep it out of the way of W_hh in L2
            dst[j] = v;
        }
    };
    auto load_w = [&](int kb, WPair (&dst)[NBW]) {
        if KRK_DBGBIT(a, 1) return;
        const __bf16* wk = wbase + (size_t)kb * kstride;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const __bf16* p = wk + (size_t)(wave + NW * j) * 1024;
            dst[j].hi = *reinterpret_cast<const bf16x8*>(p);
            dst[j].lo = *reinterpret_cast<const bf16x8*>(p + 512);
        }
    };
    auto mma_block = [&](int kb, const unsigned char* hcur, const WPair (&w)[NBW]) {
        if KRK_DBGBIT(a, 4) return;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned char* hp = hcur + (16 * g + line) * RS + (kb * 32 + us * 8) * 2;
            const bf16x8 hh = *reinterpret_cast<const bf16x8*>(hp);
            const bf16x8 hl = *reinterpret_cast<const bf16x8*>(hp + plane);
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j].hi, hh, acc[g][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j].hi, hl, acc[g][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j].lo, hh, acc[g][j], 0, 0, 0);
        }
    };

    WPair wa[NBW], wb[NBW];
    if KRK_DBGBIT(a, 1) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) { wa[j].hi = bf16x8{}; wa[j].lo = bf16x8{}; wb[j].hi = bf16x8{}; wb[j].lo = bf16x8{}; }
    }
    int cur = 0;
    // one time step; with XPRE `xbuf` holds xproj of this step and is refilled with the next step's after the last
    // weight load (vmcnt retires in order: an HBM-latency load in front of weight loads would stall their MFMAs)
    auto step = [&](int s, f32x4 (&xbuf)[XPRE ? G : 1][XPRE ? NBW : 1]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if constexpr (XPRE) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) acc[g][j] = xbuf[g][j];
            } else {
                load_x(s, g, acc[g]);
            }
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
            if (s + 1 < Lmax && !KRK_DBGBIT(a, 8)) {
#pragma unroll
                for (int g = 0; g < G; ++g) load_x(s + 1, g, xbuf[g]);
            }
        }

        // ---- gate non-linearities, cell update (per lane: one (line, unit)), h_t -> LDS as (hi, lo)
        unsigned char* hnext = hs + (cur ^ 1) * buf;
        if (!KRK_DBGBIT(a, 2))
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int unit = (wave + NW * j) * 4 + us;
            const float gi = krk_sigmoid(acc[g][j][0]);
            const float gf = krk_sigmoid(acc[g][j][1]);
            const float gg = krk_tanh(acc[g][j][2]);
            const float go = krk_sigmoid(acc[g][j][3]);
            const float c = gf * cst[g][j] + gi * gg;
            cst[g][j] = c;
            const float h = go * krk_tanh(c);
            const __bf16 hh = (__bf16)h;
            __bf16* dst = reinterpret_cast<__bf16*>(hnext + (16 * g + line) * RS) + unit;
            dst[0] = hh;
            *reinterpret_cast<__bf16*>(reinterpret_cast<unsigned char*>(dst) + plane) = (__bf16)(h - (float)hh);
        }
        __syncthreads();
        lstm_x3_store<NW, M>(a, hnext, lens_s, s, wave, lane, dir, rev, n0);
        cur ^= 1;
    };

    f32x4 xa[XPRE ? G : 1][XPRE ? NBW : 1];
    load_w(0, wa);
    if constexpr (XPRE) {
#pragma unroll
        for (int g = 0; g < G; ++g) load_x(0, g, xa[g]);
    }
    for (int s = 0; s < Lmax; ++s) step(s, xa);
}

template <int NW, int M>
__device__ __forceinline__ void lstm_x3_idle(const LstmX3Args& a, const unsigned char* hs, const int* lens_s, int Lmax,
                                             int wave, int lane, int dir, bool rev, int n0) {
    const int buf = 2 * M * a.hrow;
    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        __syncthreads();
        lstm_x3_store<NW, M>(a, hs + (cur ^ 1) * buf, lens_s, s, wave, lane, dir, rev, n0);
        cur ^= 1;
    }
}

template <int NW, int G, int MAXB, bool XPRE>
__global__ void __launch_bounds__(64 * NW, 1) lstm_x3_kernel(const LstmX3Args a) {
    constexpr int M = 16 * G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    unsigned char* hs = smem8;                                       // [2 buffers][2 planes][M][hrow]
    int* lens_s = reinterpret_cast<int*>(smem8 + 4 * M * a.hrow);   // [M]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, direction = workgroup id % ndir: the hardware deals workgroups to XCDs round-robin by id, so with two
    // directions even XCDs only ever hold the forward weights in their L2 and odd XCDs the reverse ones
    const int dir = blockIdx.x % a.ndir;
    const bool rev = (a.dirmode == 1) || (a.dirmode == 2 && dir == 1);
    const int n0 = (blockIdx.x / a.ndir) * M;

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
        lstm_x3_loop<NW, G, MAXB, XPRE>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
    } else {
        if constexpr (MAXB > 1) lstm_x3_loop<NW, G, MAXB - 1, XPRE>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
        else lstm_x3_idle<NW, M>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
    }
}

// Hidden sizes 257 ... 512 in a split-bf16 plan (round 6).  The kernel above keeps the accumulators and the cell state of ALL of a wave's
// blocks in registers (NB <= 64); above that the recurrence fell to lstm_big_kernel -- exact f32, 30 ms per layer at 512 hidden units.
// Here a wave walks its gate-column blocks FOUR at a time (block-major: each group runs its whole K loop, then its cell updates), the
// cell state lives in LDS ([unit][16 lines] floats: a lane only ever touches its own (unit, line)), the weight fragments of the next K
// block are requested while this one's twelve MFMAs run.  Same arithmetic per accumulator as lstm_x3_loop (x, then K blocks ascending,
// hi.hi + hi.lo + lo.hi per block), same LDS rows of h, same output pass: the next layer's projection reads split planes as before.
template <int NW>
__global__ void __launch_bounds__(64 * NW, 1) lstm_x3b_kernel(const LstmX3Args a) {
    constexpr int M = 16, NA = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    unsigned char* hs = smem8;                                        // [2 buffers][2 planes][M][hrow]
    float* cs = reinterpret_cast<float*>(smem8 + 4 * M * a.hrow);    // [NB * 4 units][16 lines]
    int* lens_s = reinterpret_cast<int*>(cs + (size_t)a.NB * 4 * M);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x % a.ndir;
    const bool rev = (a.dirmode == 1) || (a.dirmode == 2 && dir == 1);
    const int n0 = (blockIdx.x / a.ndir) * M;
    if (tid < M) {
        const int n = n0 + tid;
        int l = 0;
        if (n < a.N) l = a.lens ? min(max(a.lens[n], 0), a.T) : a.T;
        lens_s[tid] = l;
    }
    for (int e = tid; e < M * a.hrow + a.NB * 4 * M; e += 64 * NW) reinterpret_cast<unsigned int*>(hs)[e] = 0u;   // h (both buffers) and c
    __syncthreads();
    int Lmax = 0;
    for (int i = 0; i < M; ++i) Lmax = max(Lmax, lens_s[i]);

    const int line = lane & 15, us = lane >> 4;
    const int RS = a.hrow, plane = M * RS, buf = 2 * plane;
    const int mylen = lens_s[line];
    const __bf16* wbase = a.wp + ((size_t)dir * a.NKB * a.NB * 2 * 64 + lane) * 8;
    const size_t kstride = (size_t)a.NB * 1024;
    const float* xrow0 = a.xp + (a.xtiled ? ((size_t)n0 * a.T + line) : (size_t)(n0 + line) * a.T) * a.xstride + (size_t)dir * a.G + us * 4;
    const size_t xg_step = a.xtiled ? 16 : 1;
    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        const bool on = s < mylen;
        const int t = on ? (rev ? (mylen - 1 - s) : s) : 0;
        const float* xr = xrow0 + (size_t)t * xg_step * a.xstride;
        const unsigned char* hcur = hs + cur * buf;
        unsigned char* hnext = hs + (cur ^ 1) * buf;
        const unsigned char* hp0 = hcur + line * RS + us * 16;
        for (int b0 = wave; b0 < a.NB; b0 += NW * NA) {
            f32x4 acc[NA];
            const __bf16* wb[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int b = min(b0 + NW * j, a.NB - 1);            // (a block past the end recomputes the last one; never stored)
                wb[j] = wbase + (size_t)b * 1024;
                acc[j] = on ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + b * 16)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            WPair wa[NA], wn[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                wa[j].hi = *reinterpret_cast<const bf16x8*>(wb[j]);
                wa[j].lo = *reinterpret_cast<const bf16x8*>(wb[j] + 512);
            }
            for (int kb = 0; kb < a.NKB; ++kb) {
                const int kn = min(kb + 1, a.NKB - 1);
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    wn[j].hi = *reinterpret_cast<const bf16x8*>(wb[j] + (size_t)kn * kstride);
                    wn[j].lo = *reinterpret_cast<const bf16x8*>(wb[j] + (size_t)kn * kstride + 512);
                }
                const bf16x8 hh = *reinterpret_cast<const bf16x8*>(hp0 + kb * 64);
                const bf16x8 hl = *reinterpret_cast<const bf16x8*>(hp0 + kb * 64 + plane);
#pragma unroll
                for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j].hi, hh, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j].hi, hl, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j].lo, hh, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NA; ++j) wa[j] = wn[j];
            }
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int b = b0 + NW * j;
                if (b >= a.NB) break;
                const int unit = b * 4 + us;
                const float gi = krk_sigmoid(acc[j][0]);
                const float gf = krk_sigmoid(acc[j][1]);
                const float gg = krk_tanh(acc[j][2]);
                const float go = krk_sigmoid(acc[j][3]);
                const float c = gf * cs[unit * M + line] + gi * gg;
                cs[unit * M + line] = c;
                const float h = go * krk_tanh(c);
                const __bf16 hh = (__bf16)h;
                __bf16* dst = reinterpret_cast<__bf16*>(hnext + line * RS) + unit;
                dst[0] = hh;
                *reinterpret_cast<__bf16*>(reinterpret_cast<unsigned char*>(dst) + plane) = (__bf16)(h - (float)hh);
            }
        }
        __syncthreads();
        lstm_x3_store<NW, M>(a, hnext, lens_s, s, wave, lane, dir, rev, n0);
        cur ^= 1;
    }
}

template <int NW, int G, int MAXB, bool XPRE>
int launch_one(const LstmX3Args& a, hipStream_t s) {
    constexpr int M = 16 * G;
    dim3 grid((unsigned)((a.N + M - 1) / M * a.ndir));
    const size_t lds = (size_t)4 * M * a.hrow + M * sizeof(int);
    auto kfn = lstm_x3_kernel<NW, G, MAXB, XPRE>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(64 * NW), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// a.NB = 4*Hp/16 column blocks of 16 gate columns; a.NKB = K-blocks of 32 (Hp padded to a multiple of 32).
// Waves per workgroup: 8 (two per SIMD, <= 256 VGPRs) once a wave would own more than 4 blocks -- the second
// wave hides the L2 weight stream and the serial gate math of the first; 4 otherwise.
// Lines per workgroup: 16, or (KRK_LSTM_STREAM_G=2) 32 = two groups of 16 sharing every weight fragment: the per-CU weight
// stream is then paid once per 32 lines and the kernel holds half as many CUs (chip time per line -33 %), but a
// launch takes 1.9 instead of 1.36 ms and the pipelined engine loses more to the longer per-batch chain.
int krk_launch_lstm_x3(const LstmX3Args& a, hipStream_t s) {
    if (a.NB > 64) {                       // 257 ... 512 hidden units: the block-major kernel with the cell state in LDS
        if (a.NB > 128 || a.N <= 0) return a.N <= 0 ? 0 : -4;
        dim3 grid((unsigned)((a.N + 15) / 16 * a.ndir));
        const size_t lds = (size_t)4 * 16 * a.hrow + (size_t)a.NB * 4 * 16 * sizeof(float) + 16 * sizeof(int);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_x3b_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lstm_x3b_kernel<8>, grid, dim3(512), lds, s, a);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    int nw = a.NB > 16 && a.NB <= 64 ? 8 : 4;
    if (const char* e = getenv("KRK_LSTM_STREAM_NW")) nw = atoi(e) == 8 && a.NB <= 64 ? 8 : 4;
    int g = 1;   // 32-line tiles are opt-in: measured 80 k vs 91 k lines/s on the pipelined bench (longer per-batch chain)
    if (const char* e = getenv("KRK_LSTM_STREAM_G")) g = (atoi(e) == 2 && nw == 8 && a.NB <= 56) ? 2 : 1;
    if (nw == 8) {
        const int per_wave = (a.NB + 7) / 8;
        if (g == 2) {
#define KRK_CASE(B_) case B_: return launch_one<8, 2, B_, false>(a, s)
            switch (per_wave) {
                KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7);
                default: return -4;
            }
#undef KRK_CASE
        }
#define KRK_CASE(B_) case B_: return launch_one<8, 1, B_, true>(a, s)
        switch (per_wave) {
            KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7); KRK_CASE(8);
            default: return -4;
        }
#undef KRK_CASE
    }
    const int per_wave = (a.NB + 3) / 4;
#define KRK_CASE(B_) case B_: return launch_one<4, 1, B_, true>(a, s)
    switch (per_wave) {
        KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7); KRK_CASE(8);
        KRK_CASE(9); KRK_CASE(10); KRK_CASE(11); KRK_CASE(12); KRK_CASE(13); KRK_CASE(14); KRK_CASE(15); KRK_CASE(16);
        default: return -4;
    }
#undef KRK_CASE
}
