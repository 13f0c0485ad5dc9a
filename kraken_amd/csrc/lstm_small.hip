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
    float* orow = a.out + (size_t)min(n, a.N - 1) * a.T * a.ostride + (size_t)dir * a.H;
    auto load_x = [&](int s, f32x4 (&dst)[NB]) {
        const bool on = s < len;
        const int t = on ? (rev ? len - 1 - s : s) : 0;
        const float* xr = xrow + (size_t)t * a.xstride;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            dst[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on) dst[b] = *reinterpret_cast<const f32x4*>(xr + b * 16);
        }
    };

    f32x4 xn[NB];
    load_x(0, xn);
    for (int s = 0; s < Lmax; ++s) {
        f32x4 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = xn[b];
        if (s + 1 < Lmax) load_x(s + 1, xn);
#pragma unroll
        for (int ks = 0; ks < NB; ++ks)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[b][ks], h[ks], acc[b], 0, 0, 0);
        const bool on = s < len;
        const int t = rev ? len - 1 - s : s;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float gi = krk_sigmoid(acc[b][0]);
            const float gf = krk_sigmoid(acc[b][1]);
            const float gg = krk_tanh(acc[b][2]);
            const float go = krk_sigmoid(acc[b][3]);
            c[b] = gf * c[b] + gi * gg;
            h[b] = go * krk_tanh(c[b]);
            const int unit = 4 * b + us;
            if (on && unit < a.H) orow[(size_t)t * a.ostride + unit] = h[b];
        }
    }
}

}  // namespace

bool krk_lstm_small_supported(int Hp) { return Hp >= 4 && Hp <= 32 && Hp % 4 == 0; }

int krk_launch_lstm_small(const LstmArgs& a, hipStream_t s) {
    if (a.N <= 0 || a.T <= 0) return 0;
    dim3 grid((unsigned)((a.N + 15) / 16 * a.ndir));
#define KRK_CASE(B_) case B_: hipLaunchKernelGGL(lstm_small_kernel<B_>, grid, dim3(64), 0, s, a); break
    switch (a.Hp / 4) {
        KRK_CASE(1); KRK_CASE(2); KRK_CASE(3); KRK_CASE(4); KRK_CASE(5); KRK_CASE(6); KRK_CASE(7); KRK_CASE(8);
        default: return -4;
    }
#undef KRK_CASE
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
