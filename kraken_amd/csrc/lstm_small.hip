// Recurrent part of LSTMs with a SMALL hidden size (<= 32, f32 plan): the 2-D LSTMs of the BLLA segmenter (Lbx32 / Lby32 over
// 450 image rows or 338 image columns, kraken/configs/vgsl.py:122) and small recognisers.  Reference semantics as in
// lstm_rec.hip (kraken/lib/vgsl/layers.py:513-547, torch.nn.LSTM gates i,f,g,o).
//
// With Hp <= 32 the whole recurrent matrix of one direction is 4*Hp*Hp*4 B <= 16 KB: it fits the registers of ONE wave as
// v_mfma_f32_16x16x4_f32 A fragments (Hp/4 column blocks x Hp/4 K steps x 1 VGPR).  One wave = 16 sequences of one
// direction, all time steps, and -- the point of this kernel -- nothing is shared between lanes across steps:
//   D = W . h^T  (A = 16 gate columns x 4 K, B = 4 K x 16 lines): lane l holds D rows 4*(l>>4)+r = the four gates of
//   unit 4*b + (l>>4) of column block b for line l&15, so the cell update is per lane, and the h value it produces is
//   exactly the B-fragment element (k = 4*ks + (l>>4), line l&15) this lane must feed for K step ks = b of the next
//   time step.  h and c never leave registers: no LDS, no barrier, no cross-lane traffic; a step is Hp/4 16-byte xproj
//   loads (prefetched one step ahead), (Hp/4)^2 MFMAs and the gate math.
// The generic kernel (lstm_rec.hip) spends ~3.5 us per step of these layers on its barrier/LDS skeleton.
#include "common.h"

namespace {

template <int NB>   // NB = Hp/4 column blocks of 16 gate columns = K steps of 4
__global__ void __launch_bounds__(64) lstm_small_kernel(const LstmArgs a) {
    const int lane = threadIdx.x;
    const int line = lane & 15, us = lane >> 4;
    const int dir = blockIdx.x % a.ndir;
    const bool rev = (a.dirmode == 1) || (a.dirmode == 2 && dir == 1);
    const int n = (blockIdx.x / a.ndir) * 16 + line;
    int len = 0;
    if (n < a.N) len = a.lens ? min(max(a.lens[n], 0), a.T) : a.T;
    int Lmax = len;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) Lmax = max(Lmax, __shfl_xor(Lmax, o));

    // resident weights: wp[dir][b][ks][lane] = W[gate column 16*b + (l&15)][k = 4*ks + (l>>4)]
    float w[NB][NB];
    const float* wp = a.wp + (size_t)dir * NB * NB * 64 + lane;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int ks = 0; ks < NB; ++ks) w[b][ks] = wp[(b * NB + ks) * 64];

    float h[NB], c[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) h[b] = c[b] = 0.f;

    const float* xrow = a.xp + (size_t)min(n, a.N - 1) * a.T * a.xstride + (size_t)dir * a.G + us * 4;
    // Vector memory of the time loop is BRANCH-FREE: loads always happen (from step 0 once a line is finished: what a finished line
    // computes is never stored), stores go through a raw buffer descriptor with an out-of-range offset for lanes that have nothing
    // to store.  Under exec-mask branches the compiler cannot count what is in flight and waits with vmcnt(0) -- which, loads
    // returning in order, makes every step wait for the rows requested last, i.e. turns the prefetch distance into one step.
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)min((size_t)a.N * a.T * a.ostride * 4, (size_t)0x7FFFF000), 0x00020000);
    const unsigned obase = (unsigned)(((size_t)min(n, a.N - 1) * a.T * a.ostride + (size_t)dir * a.H + us) * 4);
    auto load_x = [&](int s, f32x4 (&dst)[NB]) {
        const int t = s < len ? (rev ? len - 1 - s : s) : 0;
        const float* xr = xrow + (size_t)t * a.xstride;
#pragma unroll
        for (int b = 0; b < NB; ++b) dst[b] = *reinterpret_cast<const f32x4*>(xr + b * 16);
    };

    // xproj rows arrive from HBM (a layer's projections are far larger than L2, every line's row lies T rows from the next line's):
    // fetched one step ahead, a step waited for its own memory latency (~2.4 us per step whatever the arithmetic cost -- round 4
    // found the step time unmoved by a better instruction schedule).  Three steps ahead now: three register sets, the time loop
    // unrolled by three so that they are addressed statically.
    f32x4 xq[3][NB];
    load_x(0, xq[0]);
    load_x(1, xq[1]);
    load_x(2, xq[2]);
    auto step = [&](int s, f32x4 (&xn)[NB]) {
        f32x4 acc[NB];
        float hn[NB];              // h of this step: every block's MFMAs still read the PREVIOUS step's h
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = xn[b];
        load_x(s + 3, xn);             // (unconditional: past the end it re-reads step 0, see above)
        const bool on = s < len;
        const int t = rev ? len - 1 - s : s;
        // the column blocks go through in PAIRS: the MFMAs of pair p + 1 are issued in front of the gate math of pair p (whose
        // accumulators are complete), so the VALU works in the matrix pipe's shadow; the cell update is the 7-transcendental form
        // of the split-bf16 kernels (common.h: 5 v_exp + 2 v_rcp instead of 5 + 5)
        auto gates = [&](int b) { hn[b] = krk_lstm_cell(acc[b], c[b]); };
#pragma unroll
        for (int b0 = 0; b0 < NB; b0 += 2) {
#pragma unroll
            for (int ks = 0; ks < NB; ++ks) {
                acc[b0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[b0][ks], h[ks], acc[b0], 0, 0, 0);
                if (b0 + 1 < NB) acc[b0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[b0 + 1][ks], h[ks], acc[b0 + 1], 0, 0, 0);
            }
            if (b0 >= 2) {
                gates(b0 - 2);
                gates(b0 - 1);
            }
        }
        gates((NB - 1) & ~1);
        if (NB > 1 && (NB & 1) == 0) gates(NB - 1);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            h[b] = hn[b];
            const int unit = 4 * b + us;
            const unsigned vo = (on && unit < a.H) ? obase + (unsigned)(((size_t)t * a.ostride + 4 * b) * 4) : 0x7FFFF000u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn[b]), ors, vo, 0, 0);
        }
    };
    for (int s = 0; s < Lmax; s += 3) {
        step(s, xq[0]);
        if (s + 1 < Lmax) step(s + 1, xq[1]);
        if (s + 2 < Lmax) step(s + 2, xq[2]);
    }
}

// The same walk on the bf16 matrix cores with split operands (bf16x3 plans: the 2-D LSTMs of the segmenter, small final LSTMs).
// At Hp <= 32 the whole K axis is ONE v_mfma_f32_16x16x32_bf16 K block, and the register trick carries over: K slot (g = lane >> 4,
// j = 0..7) of the B operand is defined to be unit 4 j + g -- exactly the unit whose cell update lane (line, g) performs for
// column block j -- and the weights are packed with the same permutation (capi.hip: pack_lstm_small_x3).  A step is
// 3 NB MFMAs (hh, hl, lh: fp32-class products) instead of NB^2 exact-f32 ones: the f32 kernel's step is bound by its 64 MFMAs
// (4096 matrix-pipe cycles at Hp = 32), this one by the gate math.
typedef __bf16 sm_bf16x8 __attribute__((ext_vector_type(8)));
template <int NB>
__global__ void __launch_bounds__(64) lstm_small_x3_kernel(const LstmArgs a, const __bf16* __restrict__ wx) {
    const int lane = threadIdx.x;
    const int line = lane & 15, us = lane >> 4;
    const int dir = blockIdx.x % a.ndir;
    const bool rev = (a.dirmode == 1) || (a.dirmode == 2 && dir == 1);
    const int n = (blockIdx.x / a.ndir) * 16 + line;
    int len = 0;
    if (n < a.N) len = a.lens ? min(max(a.lens[n], 0), a.T) : a.T;
    int Lmax = len;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) Lmax = max(Lmax, __shfl_xor(Lmax, o));

    // resident weights: wx[dir][b][plane][lane][8]: W[gate column 16 b + (l & 15)][unit 4 j + (l >> 4)], j = 0..7
    sm_bf16x8 wh[NB], wl[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        wh[b] = *reinterpret_cast<const sm_bf16x8*>(wx + ((((size_t)dir * NB + b) * 2 + 0) * 64 + lane) * 8);
        wl[b] = *reinterpret_cast<const sm_bf16x8*>(wx + ((((size_t)dir * NB + b) * 2 + 1) * 64 + lane) * 8);
    }
    float h[8], c[NB];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) c[b] = 0.f;

    const float* xrow = a.xp + (size_t)min(n, a.N - 1) * a.T * a.xstride + (size_t)dir * a.G + us * 4;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)min((size_t)a.N * a.T * a.ostride * 4, (size_t)0x7FFFF000), 0x00020000);
    const unsigned obase = (unsigned)(((size_t)min(n, a.N - 1) * a.T * a.ostride + (size_t)dir * a.H + us) * 4);
    auto load_x = [&](int s, f32x4 (&dst)[NB]) {       // branch-free, three steps ahead: see lstm_small_kernel
        const int t = s < len ? (rev ? len - 1 - s : s) : 0;
        const float* xr = xrow + (size_t)t * a.xstride;
#pragma unroll
        for (int b = 0; b < NB; ++b) dst[b] = *reinterpret_cast<const f32x4*>(xr + b * 16);
    };
    f32x4 xq[3][NB];
    load_x(0, xq[0]);
    load_x(1, xq[1]);
    load_x(2, xq[2]);
    auto step = [&](int s, f32x4 (&xn)[NB]) {
        f32x4 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = xn[b];
        load_x(s + 3, xn);
        sm_bf16x8 bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 hi = (__bf16)h[j];
            bh[j] = hi;
            bl[j] = (__bf16)(h[j] - (float)hi);
        }
        const bool on = s < len;
        const int t = rev ? len - 1 - s : s;
        // the three MFMAs of block b + 1 are issued in front of the cell update of block b (whose accumulator is complete): the step
        // is bound by the eight cell updates, the matrix work runs in their shadow.  (bh / bl hold the PREVIOUS step's h: h[] may be
        // overwritten block by block.)
        auto mma = [&](int b) {
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[b], bh, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[b], bl, acc[b], 0, 0, 0);
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[b], bh, acc[b], 0, 0, 0);
        };
        mma(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) mma(b + 1);
            const float hv = krk_lstm_cell(acc[b], c[b]);
            h[b] = hv;
            const int unit = 4 * b + us;
            const unsigned vo = (on && unit < a.H) ? obase + (unsigned)(((size_t)t * a.ostride + 4 * b) * 4) : 0x7FFFF000u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hv), ors, vo, 0, 0);
        }
    };
    for (int s = 0; s < Lmax; s += 3) {
        step(s, xq[0]);
        if (s + 1 < Lmax) step(s + 1, xq[1]);
        if (s + 2 < Lmax) step(s + 2, xq[2]);
    }
}

}  // namespace

bool krk_lstm_small_supported(int Hp) { return Hp >= 4 && Hp <= 32 && Hp % 4 == 0; }

int krk_launch_lstm_small_x3(const LstmArgs& a, const void* wx, hipStream_t s) {
    if (a.N <= 0 || a.T <= 0) return 0;
    if ((size_t)a.N * a.T * a.ostride * 4 >= (size_t)0x7FFFF000) return -4;
    dim3 grid((unsigned)((a.N + 15) / 16 * a.ndir));
#define KRK_CASE(B_) case B_: hipLaunchKernelGGL(lstm_small_x3_kernel<B_>, grid, dim3(64), 0, s, a, (const __bf16*)wx); break
    switch (a.Hp / 4) {
        KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7); KRK_CASE(8);
        default: return -4;
    }
#undef KRK_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int krk_launch_lstm_small(const LstmArgs& a, hipStream_t s) {
    if (a.N <= 0 || a.T <= 0) return 0;
    if ((size_t)a.N * a.T * a.ostride * 4 >= (size_t)0x7FFFF000) return -4;     // 32-bit store offsets: the generic kernel takes it
    dim3 grid((unsigned)((a.N + 15) / 16 * a.ndir));
#define KRK_CASE(B_) case B_: hipLaunchKernelGGL(lstm_small_kernel<B_>, grid, dim3(64), 0, s, a); break
    switch (a.Hp / 4) {
        KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7); KRK_CASE(8);
        default: return -4;
    }
#undef KRK_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
