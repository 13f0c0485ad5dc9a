// Wide-kernel convolution (11 <= kw <= 16, stride 1) on the gfx950 bf16 matrix cores with split operands,
// using the horizontal TAPS of one input channel as the MFMA K block (the scheme of conv1_x3.hip extended to
// Cin channels).  Reference: kraken/lib/vgsl/layers.py ActConv2D.forward :842-860 (+ fused MaxPool :381-388);
// this is the 3x13 second convolution of kraken's default recognition spec.
//
// Why: with channels as K (conv_x3.hip) every (tap, pixel fragment) is re-read from LDS -- 39 taps x 2 KB per
// 3 MFMAs -- and every wave streams the whole filter bank per tile; the kernel sat at 40 % MFMA busy.  With
// taps as K one 12-pixel window read per (channel, input row) feeds 2 output rows x 2 segments x 3 terms = 12
// MFMAs, and a 2 KB weight fragment pair (channel, kernel row) is used for 12 MFMAs too: 3x less LDS and 4x
// less L1 traffic per MFMA, for 16/13 more MFMA work (taps 13..15 carry zero weights).
//
//   input     split bf16 planes in "NHCW" order [N][H][Cin][pitch] (pitch % 8 == 0, written by conv1_x3.hip)
//   tile      4 waves = 2 row pairs x 2 column halves; a wave owns 2 output rows x 64 columns, its 32 MFMA
//             columns are pixels 2c + s (s = 0, 1): lane c reads pixels 2c + 8*half + s + shift ..+7
//   LDS       [plane][4 channels][kh+3 rows][152 columns], two buffers, register-staged one chunk ahead
//   weights   [Cin][kh][plane][lane][8] A fragments straight from L2, prefetched one channel ahead
//   epilogue  2x2 max-pool inside a lane, bias + activation, length mask, split NHWC planes (for conv_x3.hip)
//
// FIVE (kw <= 13, the split-operand build only): 13 taps x 3 product terms = 39 K slots, which the scheme above spreads over
// 3 MFMAs of 16 (taps 13..15 idle in each: 16/13 of the algorithmic MFMA work).  The slot <-> (term, tap) assignment is free as
// long as the weights are packed to match, so the 39 slots of one (channel, kernel row) are packed into FIVE groups of 8 and the
// groups of TWO channels share an MFMA (lane half = channel): 5 MFMAs per channel pair instead of 6.
//     group   B operand (8 consecutive K slots of a lane)                     A operand
//     Ga      hi pixels 0..7                                                  w_hi taps 0..7   (hi x hi)
//     Ga      (the same registers)                                            w_lo taps 0..7   (lo x hi)
//     Gc      lo pixels 0..7                                                  w_hi taps 0..7   (hi x lo: the fragment of row 1)
//     Gd      hi pixels 8..12 | lo pixels 8..10                               w_hi taps 8..12 | w_hi taps 8..10
//     Ge      lo pixels 11..12 | hi pixels 8..12 | (pad)                      w_hi taps 11..12 | w_lo taps 8..12 | 0
// A lane reads ONE 16-pixel window per plane of ITS channel (half as much LDS traffic as above) and assembles the mixed groups
// with v_alignbyte / v_perm; the 4 distinct weight fragments of a kernel row are refreshed in place for the next channel pair
// as soon as the row that used them last is done (no second register set).
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

// LDS-DMA by hand (the lstm_ws.hip idiom).  Through the builtins the compiler books every copy as a store to "some" LDS address and
// waits for it in front of the next ds_read that might alias -- the fragment reads of the CURRENT chunk: progressive vmcnt(3..0)
// waits a third of the way into a chunk's MFMAs for copies that only the NEXT chunk reads.  In asm it neither sees nor counts them;
// its own counted waits stay safe (vmcnt retires in order: a wait that covers one of its loads covers every older copy too), and the
// chunk boundary waits explicitly (vmcnt(0) + barrier).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)(size_t)(lds_ptr_t)p; }
__device__ __forceinline__ void dma_global_b128(const void* gptr, unsigned lds_off) {     // 64 lanes x 16 B -> LDS [lds_off, + 1 KB)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void dma_buffer_b128(unsigned voff, const i32x4_t& srd, unsigned soff, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(soff), "s"(lds_off) : "memory");
}

constexpr int LW6 = 152;  // LDS row of the six-group kernel: 8 (aligned left margin) + 128 + 15, rounded to 16-byte pieces
constexpr int LW5 = 144;  // five-group kernel: 8 + 128 + 8 (its windows end at column 143)
constexpr int WPAIR = 3 * 4 * 64 * 16;   // bytes of the weight fragments of one channel pair (five-group kernel)
constexpr int CC = 4;     // channels per LDS chunk
constexpr int TW = 128, TH = 4;

// 8 consecutive bf16 starting at element T of a window held in dwords d[]
template <int T>
__device__ __forceinline__ bf16x8 window8(const unsigned (&d)[6]) {
    if constexpr (T % 2 == 0) {
        return __builtin_bit_cast(bf16x8, u32x4{d[T / 2], d[T / 2 + 1], d[T / 2 + 2], d[T / 2 + 3]});
    } else {
        constexpr int b = T / 2;
        return __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbyte(d[b + 1], d[b], 2), __builtin_amdgcn_alignbyte(d[b + 2], d[b + 1], 2),
                                                __builtin_amdgcn_alignbyte(d[b + 3], d[b + 2], 2), __builtin_amdgcn_alignbyte(d[b + 4], d[b + 3], 2)});
    }
}

// ---- FIVE: pieces of a window held as dwords W[k] = pixels (2k, 2k+1)
template <int J, int NW>
__device__ __forceinline__ unsigned px2(const unsigned (&W)[NW]) {             // pixels (J, J+1)
    if constexpr (J % 2 == 0) return W[J / 2];
    else return __builtin_amdgcn_alignbyte(W[J / 2 + 1], W[J / 2], 2);
}
template <int JA, int JB, int NW>
__device__ __forceinline__ unsigned mix2(const unsigned (&A)[NW], const unsigned (&B)[NW]) {   // (pixel JA of A, pixel JB of B)
    constexpr unsigned lo_sel = (JA % 2 == 0) ? 0x0100u : 0x0302u;             // v_perm_b32: selector bytes 0..3 = src1, 4..7 = src0
    constexpr unsigned hi_sel = (JB % 2 == 0) ? 0x0504u : 0x0706u;
    return __builtin_amdgcn_perm(B[JB / 2], A[JA / 2], (hi_sel << 16) | lo_sel);
}
template <int J, int NW>
__device__ __forceinline__ unsigned px1(const unsigned (&W)[NW]) {             // (pixel J, some finite value: its weight is zero)
    if constexpr (J % 2 == 0) return W[J / 2];
    else return W[J / 2] >> 16;
}

template <int KH, int PW, bool POOL, bool FIVE, bool DMA>
__global__ void __launch_bounds__(256, 2) conv_taps_kernel(const ConvTapArgs a) {
    constexpr int IH = TH + KH - 1;
    constexpr int SHIFT = 8 - PW;                    // LDS column 0 is image column w0 - 8
    constexpr int LW = FIVE ? LW5 : LW6;
    constexpr int ROWS = 2 * CC * IH;                // LDS rows per chunk (both planes)
    constexpr int PIECES = ROWS * (LW / 8);          // 16-byte pieces per chunk
    constexpr int NST = (PIECES + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_taps[];
    auto tile = reinterpret_cast<__bf16 (*)[2][CC][IH][LW]>(smem_taps);           // [buffer 2][plane 2][CC][IH][LW]
    [[maybe_unused]] unsigned char* wts = smem_taps + sizeof(__bf16) * 2 * 2 * CC * IH * LW;   // FIVE: [chunk buffer 2][pair 2][WPAIR]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, c = lane & 31;
    const int rp = wave >> 1, chalf = wave & 1;

    int bt = blockIdx.x;
    const int tw = bt % a.tiles_w;
    bt /= a.tiles_w;
    const int th = bt % a.tiles_h;
    const int n = bt / a.tiles_h;
    const int h0 = th * TH, w0 = tw * TW;
    const int len_out = a.len_out ? a.len_out[n] : a.Wy;
    const int wlim = POOL ? min(a.Wo, 2 * len_out) : min(a.Wo, len_out);

    // ---- staging bookkeeping: piece e = tid + 256*i -> (plane, channel, row, 16-byte piece)
    long s_src[NST];
    int s_dst[NST];
    bool s_ok[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i;
        const int row = e / (LW / 8), q = e - row * (LW / 8);
        const int plane = row / (CC * IH), r2 = row - plane * (CC * IH);
        const int ch = r2 / IH, ih = r2 - ch * IH;
        const int gh = h0 - a.ph + ih, gw = w0 - 8 + 8 * q;
        s_ok[i] = e < PIECES && gh >= 0 && gh < a.H && gw >= 0 && gw < a.pitch;
        s_src[i] = (long)plane * (long)a.x_plane + (((long)n * a.H + gh) * a.Cin + ch) * a.pitch + gw;
        s_dst[i] = e * 8;
    }
    // ---- round 4: the same pieces as raw-buffer -> LDS copies (buffer_load_dwordx4 ... lds): piece e of the chunk lands at LDS
    // byte 16 e of the tile buffer (= wave-uniform base + 16 * lane), out-of-range pieces (image border, pitch) deliver zeros, lanes
    // past the tile are masked off by EXEC.  No staging registers, no ds_write pass, no exec-masked global loads whose count the
    // compiler cannot track; the chunk's channel offset travels in soffset.  One descriptor spans both planes of the line.
    constexpr bool dma = DMA;   // compile-time: with both staging forms behind a run-time flag the compiler merged their control flow
                                // and waited vmcnt(0) for the copies right where they are issued
    unsigned d_vo[NST];
    const size_t line_elems = (size_t)a.H * a.Cin * a.pitch;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i;
        const int row = e / (LW / 8), q = e - row * (LW / 8);
        const int plane = row / (CC * IH), r2 = row - plane * (CC * IH);
        const int ch = r2 / IH, ih = r2 - ch * IH;
        const int gh = h0 - a.ph + ih, gw = w0 - 8 + 8 * q;
        const bool ok = e < PIECES && gh >= 0 && gh < a.H && gw >= 0 && gw < a.pitch && !KRK_DBGBIT(a, 4);
        d_vo[i] = ok ? (unsigned)(((size_t)plane * a.x_plane + ((size_t)gh * a.Cin + ch) * a.pitch + gw) * 2) : 0xFFFFFFF0u;
    }
    i32x4_t xrs;
    {
        const unsigned long long u = reinterpret_cast<unsigned long long>(a.x + (size_t)n * line_elems);
        xrs[0] = (int)__builtin_amdgcn_readfirstlane((unsigned)u);
        xrs[1] = (int)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xFFFFu);
        xrs[2] = (int)__builtin_amdgcn_readfirstlane((unsigned)((a.x_plane + line_elems) * 2));
        xrs[3] = 0x00020000;
    }
    const unsigned tile_lds = lds_offset_of(&tile[0][0][0][0][0]);
    constexpr unsigned TILE_BYTES = sizeof(__bf16) * 2 * CC * IH * LW;
    [[maybe_unused]] auto dma_stage = [&](int ch0, int buf) {
        const unsigned t = tile_lds + (unsigned)buf * TILE_BYTES + (unsigned)wave * 1024u;
        const unsigned so = (unsigned)(ch0 * a.pitch * 2);
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if (tid + 256 * i < PIECES) dma_buffer_b128(d_vo[i], xrs, so, t + i * 4096);
    };
    [[maybe_unused]] f32x4 st[NST];
    [[maybe_unused]] auto gload = [&](int ch0) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            st[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s_ok[i] && !KRK_DBGBIT(a, 4)) st[i] = *reinterpret_cast<const f32x4*>(a.x + s_src[i] + (long)ch0 * a.pitch);
        }
    };
    [[maybe_unused]] auto lstore = [&](int buf) {
        __bf16* t = &tile[buf][0][0][0][0];
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if (tid + 256 * i < PIECES) *reinterpret_cast<f32x4*>(t + s_dst[i]) = st[i];
    };

    // ---- weights: fragments of one channel = KH kernel rows x (hi, lo), prefetched one channel ahead
    const bf16x8* wbase = reinterpret_cast<const bf16x8*>(a.wpack) + lane;
    auto wload = [&](int chg, bf16x8 (&wh)[KH], bf16x8 (&wl)[KH]) {
#pragma unroll
        for (int dy = 0; dy < KH; ++dy) {
            wh[dy] = wbase[((size_t)(chg * KH + dy) * 2 + 0) * 64];
            wl[dy] = wbase[((size_t)(chg * KH + dy) * 2 + 1) * 64];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][s][r] = 0.f;

    const bool live = (w0 + 64 * chalf) < wlim;      // this wave's columns are inside the line
    const int nchunks = a.Cin / CC;

    if constexpr (dma) {
        dma_stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        gload(0);
        lstore(0);
    }
    if constexpr (!FIVE) {
    bf16x8 wha[KH], wla[KH], whb[KH], wlb[KH];
    wload(0, wha, wla);
    __syncthreads();

    // one channel: KH+1 input rows, each window feeds both output rows of the pair
    auto channel = [&](int buf, int ch, const bf16x8 (&wh)[KH], const bf16x8 (&wl)[KH]) {
#pragma unroll
        for (int i = 0; i < KH + 1; ++i) {
            unsigned dh[6], dl[6];
            const unsigned* ph = reinterpret_cast<const unsigned*>(&tile[buf][0][ch][2 * rp + i][64 * chalf + 2 * c + 8 * half]);
            const unsigned* pl = reinterpret_cast<const unsigned*>(&tile[buf][1][ch][2 * rp + i][64 * chalf + 2 * c + 8 * half]);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                dh[k] = ph[k];
                dl[k] = pl[k];
            }
            const bf16x8 fh0 = window8<SHIFT>(dh), fh1 = window8<SHIFT + 1>(dh);
            const bf16x8 fl0 = window8<SHIFT>(dl), fl1 = window8<SHIFT + 1>(dl);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int dy = i - o;
                if (dy < 0 || dy >= KH) continue;
                acc[o][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[dy], fh0, acc[o][0], 0, 0, 0);
                acc[o][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[dy], fh1, acc[o][1], 0, 0, 0);
                KRK_CROSS(acc[o][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[dy], fl0, acc[o][0], 0, 0, 0);
                          acc[o][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[dy], fl1, acc[o][1], 0, 0, 0);
                          acc[o][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[dy], fh0, acc[o][0], 0, 0, 0);
                          acc[o][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[dy], fh1, acc[o][1], 0, 0, 0);)
            }
        }
    };

    for (int k = 0; k < nchunks; ++k) {
        const int buf = k & 1;
        if (k + 1 < nchunks) {
            if constexpr (dma) dma_stage((k + 1) * CC, buf ^ 1);
            else gload((k + 1) * CC);
        }
        if (live && !KRK_DBGBIT(a, 1)) {
            static_assert(CC == 4, "channel loop is unrolled for 4-channel chunks");
            const int cg = k * CC;
            wload(cg + 1, whb, wlb);
            channel(buf, 0, wha, wla);
            wload(cg + 2, wha, wla);
            channel(buf, 1, whb, wlb);
            wload(cg + 3, whb, wlb);
            channel(buf, 2, wha, wla);
            wload(min(cg + 4, a.Cin - 1), wha, wla);
            channel(buf, 3, whb, wlb);
        }
        if constexpr (!dma) { if (k + 1 < nchunks) lstore(buf ^ 1); }
        if constexpr (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    } else {
#ifndef KRK_BF16_ONE
    // ---- FIVE: see the header.  wq[dy][f]: f = 0 w_hi taps 0..7, 1 w_lo taps 0..7, 2 the Gd weights, 3 the Ge weights
    static_assert(KH == 3 && CC == 4 && (SHIFT == 2 || SHIFT == 3), "FIVE: 3 kernel rows, 2 channel pairs per chunk, kw 11..13");
    constexpr int NW = SHIFT == 2 ? 8 : 9;
    // The four waves of a workgroup use the SAME weight fragments: fetched once per workgroup into LDS by asynchronous
    // global -> LDS copies, one chunk ahead (per-wave loads of 24 KB per chunk ran the CU's 64 B/clk vector-memory path
    // at ~80 %, in order with the tile staging loads: 0.59 ms without the weight traffic against 0.75 ms with it)
    const int npairs = a.Cin >> 1;
    const unsigned wts_lds = lds_offset_of(wts);
    auto wdma = [&](int chunk, int wb) {       // this wave's quarter of the chunk's (two channel pairs) fragments -> weight buffer wb
        if (KRK_DBGBIT(a, 2)) return;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.wpack5) + (size_t)chunk * (2 * WPAIR) + wave * (WPAIR / 2) + lane * 16;
        const unsigned dst = wts_lds + (unsigned)(wb * (2 * WPAIR) + wave * (WPAIR / 2));
#pragma unroll
        for (int j = 0; j < WPAIR / 2 / 1024; ++j) dma_global_b128(src + j * 1024, dst + j * 1024);
    };
    auto landed = [&]() {                       // my copies are in LDS; after the barrier: everyone's, and the other buffers are free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    wdma(0, 0);
    landed();

    // one channel pair (this lane: channel 2*pp + half of the chunk): KH+1 input rows
    auto pair_rows = [&](int buf, int pp, int wb) {
        const int ch = 2 * pp + half;
        bf16x8 wq[KH][4];
        {
            const bf16x8* wl = reinterpret_cast<const bf16x8*>(wts + wb * WPAIR) + lane;      // wb = 2 * (chunk buffer) + pair of the chunk
#pragma unroll
            for (int dy = 0; dy < KH; ++dy)
#pragma unroll
                for (int f = 0; f < 4; ++f) wq[dy][f] = wl[(dy * 4 + f) * 64];
        }
#pragma unroll
        for (int i = 0; i < KH + 1; ++i) {
            unsigned Wh[NW], Wl[NW];
            const unsigned* ph = reinterpret_cast<const unsigned*>(&tile[buf][0][ch][2 * rp + i][64 * chalf + 2 * c]);
            const unsigned* pl = reinterpret_cast<const unsigned*>(&tile[buf][1][ch][2 * rp + i][64 * chalf + 2 * c]);
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                Wh[k] = ph[k];
                Wl[k] = pl[k];
            }
            auto groups = [&](auto R, bf16x8& Ga, bf16x8& Gc, bf16x8& Gd, bf16x8& Ge) {
                constexpr int r = decltype(R)::value;        // window pixel of tap 0
                Ga = __builtin_bit_cast(bf16x8, u32x4{px2<r, NW>(Wh), px2<r + 2, NW>(Wh), px2<r + 4, NW>(Wh), px2<r + 6, NW>(Wh)});
                Gc = __builtin_bit_cast(bf16x8, u32x4{px2<r, NW>(Wl), px2<r + 2, NW>(Wl), px2<r + 4, NW>(Wl), px2<r + 6, NW>(Wl)});
                Gd = __builtin_bit_cast(bf16x8, u32x4{px2<r + 8, NW>(Wh), px2<r + 10, NW>(Wh), mix2<r + 12, r + 8, NW>(Wh, Wl), px2<r + 9, NW>(Wl)});
                Ge = __builtin_bit_cast(bf16x8, u32x4{px2<r + 11, NW>(Wl), px2<r + 8, NW>(Wh), px2<r + 10, NW>(Wh), px1<r + 12, NW>(Wh)});
            };
            bf16x8 Ga[2], Gc[2], Gd[2], Ge[2];
            groups(std::integral_constant<int, SHIFT>{}, Ga[0], Gc[0], Gd[0], Ge[0]);
            groups(std::integral_constant<int, SHIFT + 1>{}, Ga[1], Gc[1], Gd[1], Ge[1]);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int dy = i - o;
                if (dy < 0 || dy >= KH) continue;
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    acc[o][sx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[dy][0], Ga[sx], acc[o][sx], 0, 0, 0);
                    acc[o][sx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[dy][1], Ga[sx], acc[o][sx], 0, 0, 0);
                    acc[o][sx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[dy][0], Gc[sx], acc[o][sx], 0, 0, 0);
                    acc[o][sx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[dy][2], Gd[sx], acc[o][sx], 0, 0, 0);
                    acc[o][sx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[dy][3], Ge[sx], acc[o][sx], 0, 0, 0);
                }
            }
        }
    };

    const bool work = live && !KRK_DBGBIT(a, 1);
    for (int k = 0; k < nchunks; ++k) {
        const int buf = k & 1;
        // everything the NEXT chunk needs is requested now and waited for at the end of this chunk: a whole chunk of MFMAs
        // (~3.5 us) covers the HBM latency of the tile loads and the copies of the weight fragments alike
        if (k + 1 < nchunks) {
            wdma(k + 1, buf ^ 1);
            if constexpr (dma) dma_stage((k + 1) * CC, buf ^ 1);
            else gload((k + 1) * CC);
        }
        if (work) {
            pair_rows(buf, 0, 2 * buf);
            pair_rows(buf, 1, 2 * buf + 1);
        }
        if constexpr (!dma) { if (k + 1 < nchunks) lstore(buf ^ 1); }
        landed();
    }
#endif
    }

    // ---- epilogue: lane = pixels w0 + 64*chalf + 2c + s of rows h0 + 2*rp + o; register 4j+i = filter 8j + 4*half + i
    f32x4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = *reinterpret_cast<const f32x4*>(a.bias + 8 * j + 4 * half);
    __bf16* yh = a.y;
    __bf16* yl = a.y + a.y_plane;
    // the activation is chosen once per tile, not per element: ReLU is a single v_max
    auto store_tile = [&](auto actf) {
    if (POOL) {
        const int prow = (h0 >> 1) + rp;
        const int pcol = (w0 >> 1) + 32 * chalf + c;
        const bool ok = prow < a.Hy && pcol < a.Wy;
        const size_t base = (size_t)n * a.y_sn + (size_t)prow * a.y_sr + (size_t)pcol * a.y_sc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 hv, lv;
                    f32x4 fv;   // the same four values unsplit, for a GroupNorm consumer (y_f32)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * j + i;
                float v = fmaxf(fmaxf(acc[0][0][r], acc[0][1][r]), fmaxf(acc[1][0][r], acc[1][1][r]));
                v = actf(v + bias4[j][i]);
                if (pcol >= len_out) v = 0.f;
                const __bf16 h = (__bf16)v;
                hv[i] = h;
                        fv[i] = v;
                lv[i] = (__bf16)(v - (float)h);
            }
            const int co = 8 * j + 4 * half;
            if (ok && co < a.Cout && !KRK_DBGBIT(a, 16)) {
                if (a.y_f32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + base + co) = fv;
                } else {
                    *reinterpret_cast<bf16x4*>(yh + base + co) = hv;
                    *reinterpret_cast<bf16x4*>(yl + base + co) = lv;
                }
            }
        }
    } else {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int row = h0 + 2 * rp + o;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int col = w0 + 64 * chalf + 2 * c + s;
                const bool ok = row < a.Ho && col < a.Wo;
                const size_t base = (size_t)n * a.y_sn + (size_t)row * a.y_sr + (size_t)col * a.y_sc;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x4 hv, lv;
                    f32x4 fv;   // the same four values unsplit, for a GroupNorm consumer (y_f32)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = actf(acc[o][s][4 * j + i] + bias4[j][i]);
                        if (col >= len_out) v = 0.f;
                        const __bf16 h = (__bf16)v;
                        hv[i] = h;
                        fv[i] = v;
                        lv[i] = (__bf16)(v - (float)h);
                    }
                    const int co = 8 * j + 4 * half;
                    if (ok && co < a.Cout && !KRK_DBGBIT(a, 16)) {
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
    };
    if (a.act == ACT_RELU) store_tile([](float v) { return fmaxf(v, 0.f); });
    else store_tile([&](float v) { return krk_act(v, a.act); });
}

template <int PW>
int launch_pw(const ConvTapArgs& a, bool pool, hipStream_t s) {
    dim3 grid((unsigned)(a.N * a.tiles_h * a.tiles_w));
    constexpr size_t tile6 = sizeof(__bf16) * 2 * 2 * CC * (TH + 2) * LW6;
#ifndef KRK_BF16_ONE
    if constexpr (PW == 5 || PW == 6) {
        if (a.wpack5 && a.kw <= 13) {
            constexpr size_t lds5 = sizeof(__bf16) * 2 * 2 * CC * (TH + 2) * LW5 + 4 * WPAIR;     // 27 KB of tiles + 48 KB of weights: two workgroups per CU
            // more than the 64 KB a kernel gets by default: raise the limit once per device (the attribute belongs to the
            // function object of the CURRENT device)
            static bool attr_set[64] = {false};
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev < 0 || dev >= 64 || !attr_set[dev]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_taps_kernel<3, PW, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_taps_kernel<3, PW, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_taps_kernel<3, PW, true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_taps_kernel<3, PW, false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
                if (dev >= 0 && dev < 64) attr_set[dev] = true;
            }
            if (a.dma) {
                if (pool) hipLaunchKernelGGL((conv_taps_kernel<3, PW, true, true, true>), grid, dim3(256), lds5, s, a);
                else hipLaunchKernelGGL((conv_taps_kernel<3, PW, false, true, true>), grid, dim3(256), lds5, s, a);
            } else {
                if (pool) hipLaunchKernelGGL((conv_taps_kernel<3, PW, true, true, false>), grid, dim3(256), lds5, s, a);
                else hipLaunchKernelGGL((conv_taps_kernel<3, PW, false, true, false>), grid, dim3(256), lds5, s, a);
            }
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
    }
#endif
    if (a.dma) {
        if (pool) hipLaunchKernelGGL((conv_taps_kernel<3, PW, true, false, true>), grid, dim3(256), tile6, s, a);
        else hipLaunchKernelGGL((conv_taps_kernel<3, PW, false, false, true>), grid, dim3(256), tile6, s, a);
    } else {
        if (pool) hipLaunchKernelGGL((conv_taps_kernel<3, PW, true, false, false>), grid, dim3(256), tile6, s, a);
        else hipLaunchKernelGGL((conv_taps_kernel<3, PW, false, false, false>), grid, dim3(256), tile6, s, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

#ifndef KRK_BF16_ONE
bool krk_conv_taps_supported(int Cin, int Cout, int kh, int kw, int sh, int sw, int dh, int dw) {
    return Cin % 4 == 0 && Cin >= 4 && Cout <= 32 && Cout % 4 == 0 && kh == 3 && kw >= 11 && kw <= 16 && sh == 1 && sw == 1 &&
           dh == 1 && dw == 1;
}

#endif

int KRK_FN(krk_launch_conv_taps)(const ConvTapArgs& a, bool pool, hipStream_t s) {
    if (a.N <= 0) return 0;
    switch (a.pw) {
        case 5: return launch_pw<5>(a, pool, s);
        case 6: return launch_pw<6>(a, pool, s);
        case 7: return launch_pw<7>(a, pool, s);
        default: return -1;
    }
}
