// Recurrent part of the (bi)directional LSTM, split-bf16 operands on the bf16 matrix cores -- second generation.
// Reference semantics: nn.LSTM inside TransposedSummarizingRNN.forward (kraken/lib/vgsl/layers.py:513-547): packed by
// length, gates i,f,g,o, h/c start at zero, outputs past a line's length stay zero.  Per step
//   gates = xproj[t] + W_hi.h_hi + W_hi.h_lo + W_lo.h_hi          (three fp32 accumulators, summed in a fixed order)
//
// Why a second generation: the first kernel (lstm_x3.hip) ran a step as a serial chain -- weight stream, MFMAs, xproj
// fetch, gate math, barrier, output pass all added up (9.1 us per step for 16 lines per CU, matrix pipe 20 % busy).
// This one keeps the matrix pipe fed:
//   * one workgroup = (NT tiles of 16 lines, one direction); every weight fragment a wave pulls from L2 feeds 3*NT MFMAs,
//     so the per-CU weight stream (the bytes of fp32 W_hh per step: 0.7 MB for H = 200) is paid once per 16*NT lines;
//   * h_{t-1} is REGISTER resident for the whole step (A/B fragments of all K blocks: 8*NT*NKB registers), loaded once
//     per step from LDS; the loop order is gate-column-block outer, K inner, so only 3*NT accumulators are live and the
//     gate math of block j (VALU / transcendental) sits between the MFMAs of block j+1 instead of after all of them;
//   * weights stream in the order they are used, ([block][kb] fragment order, see upload_lstm_x3v2) through a register ring
//     D stages deep that runs continuously across blocks AND across time steps (buffer loads: scalar offsets, no VALU);
//   * xproj comes from HBM and vmcnt retires in order, so an xproj load in front of a weight load stalls that weight's
//     MFMAs for an HBM latency.  Every wave therefore fetches xproj in TWO bursts per step, each refilling the entries its
//     blocks consumed since the previous burst (needed again half a step later), and the two waves that share a SIMD
//     (w, w+4) burst a quarter step apart: while one waits the other has the matrix pipe to itself;
//   * h_t goes to LDS as (hi, lo) rows (the next step's B operand); after the step barrier a branch-free pass copies it to
//     the K-blocked split planes the next projection (gemm_x3.hip) streams, 16 bytes per lane, masked by the hardware
//     bounds check of the buffer descriptor (out-of-range offset = dropped store) -- it overlaps the first MFMAs.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {


constexpr unsigned kOOB = 0x80000000u;   // voffset beyond any descriptor here (launcher keeps them < 2 GiB; no 32-bit wrap with a folded immediate): load returns 0, store is dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    // base and size are wave-uniform by construction; readfirstlane makes that provable (no waterfall loops)
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

struct WFrag {
    u32x4 hi, lo;
};

__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) {
    return __builtin_bit_cast(bf16x8, v);
}

// K blocks whose LO h fragments stay in registers for the whole step (the HI fragments always do); the rest are read from
// LDS where they are used -- the register budget (256 per lane with two waves per SIMD, 512 with one) decides
constexpr int hl_resident(int NW, int NT, int NKB) {
    if (NW == 8) return NT == 1 ? NKB : (NKB <= 6 ? NKB : (NKB == 7 ? 5 : 1));
    return NT <= 2 ? NKB : 0;
}

// NW waves, NT line tiles of 16, NKB K blocks of 32, NBW gate-column blocks for this wave, (P1, P2) xproj burst positions
template <int NW, int NT, int NKB, int NBW, int P1, int P2>
__device__ __forceinline__ void lstm_v2_loop(const LstmX3Args& a, unsigned char* hs, const int* lens_s, int Lmax,
                                             int wave, int lane, int dir, bool rev, int n0) {
    constexpr int M = 16 * NT;
    constexpr int D = 3;                      // weight ring depth (stages of one (block, kb) fragment pair)
    constexpr int Q = NBW * NKB;              // stages per step
    static_assert(P1 >= 0 && P1 < P2 && P2 <= NBW, "burst positions");
    const int line = lane & 15;
    const int us = lane >> 4;
    const int RS = a.hrow;
    const int plane = M * RS;
    const int buf = 2 * plane;

    int mylen[NT];
#pragma unroll
    for (int g = 0; g < NT; ++g) mylen[g] = lens_s[16 * g + line];

    // ---- descriptors
    const size_t wdir = (size_t)a.NB * NKB * 1024;                            // bf16 elements per direction
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.wp + (size_t)dir * wdir, (unsigned)(wdir * 2));
    const unsigned wvo = lane * 16;
    unsigned wso0 = (unsigned)wave * NKB * 2048;                              // block `wave`, scalar
    const int nrows = min(a.N - n0, M);
    const size_t xrow_bytes = (size_t)a.xstride * 4;
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.xp + (size_t)n0 * a.T * a.xstride + (size_t)dir * a.G,
                                                 (unsigned)((size_t)nrows * a.T * xrow_bytes - (size_t)dir * a.G * 4));
    unsigned xso0 = (unsigned)wave * 64;                                      // block `wave`: 16 gate columns x 4 bytes

    auto xvoff = [&](int s, int g) -> unsigned {
        const bool on = s < mylen[g];
        const int t = rev ? (mylen[g] - 1 - s) : s;
        return on ? (unsigned)(((unsigned)(16 * g + line) * (unsigned)a.T + (unsigned)t) * (unsigned)xrow_bytes + us * 16) : kOOB;
    };
    auto load_w = [&](int q, WFrag& dst) {     // q = j * NKB + kb of THIS wave's stage sequence
        if (KRK_DBGBIT(a, 1)) return;
        const int j = q / NKB, kb = q - j * NKB;
        const unsigned so = wso0 + (unsigned)((NW * j * NKB + kb) * 2048);
        dst.hi = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0);
        dst.lo = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so + 1024, 0);
    };

    // ---- state
    float cst[NBW][NT];
    f32x4 xr[NBW][NT];
    WFrag ring[D];
    constexpr int HLR = hl_resident(NW, NT, NKB);
    bf16x8 hh[NKB][NT], hl[HLR > 0 ? HLR : 1][NT];
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int g = 0; g < NT; ++g) cst[j][g] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { ring[d].hi = u32x4{0, 0, 0, 0}; ring[d].lo = u32x4{0, 0, 0, 0}; }

    auto load_x = [&](int j, const unsigned (&vo)[NT]) {
        if (KRK_DBGBIT(a, 8)) return;
#pragma unroll
        for (int g = 0; g < NT; ++g)
            xr[j][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo[g], xso0 + (unsigned)(NW * j * 64), 2 /* nt: read once */));
    };

    // ---- output pass descriptors: this wave copies LPW lines x (H/8 pieces) x 2 planes of h_{s-1} per step
    constexpr int LPW = (M + NW - 1) / NW;
    constexpr int NPC = (LPW * NKB * 4 * 2 + 63) / 64;
    const int per_line = max(a.H >> 3, 1);
    const size_t rows_total = (size_t)a.N * a.T;
    const __amdgpu_buffer_rsrc_t ors = make_rsrc(a.out, (unsigned)((size_t)a.out_plane * 4));   // both planes (launcher checks < 4 GB)
    unsigned p_lds[NPC], p_g0[NPC];
    int p_len[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int e = lane + 64 * i;
        const int total = LPW * per_line * 2;
        const int pl = e / (LPW * per_line), r = e - pl * LPW * per_line;
        const int li = r / per_line, q = r - li * per_line;
        const int ln = wave * LPW + li;
        const bool ok = e < total && ln < M;
        p_lds[i] = ok ? (unsigned)(pl * plane + ln * RS + q * 16) : 0u;
        p_g0[i] = (unsigned)((((size_t)(dir * per_line + q)) * rows_total + (size_t)(n0 + ln) * a.T) * 16 + (size_t)pl * a.out_plane * 2);
        p_len[i] = ok ? lens_s[ln] : 0;
    }
    auto store_pass = [&](int s, const unsigned char* hb) {     // h of step s, sitting in LDS buffer hb
        if (KRK_DBGBIT(a, 16)) return;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const bool on = s < p_len[i];
            const int t = rev ? (p_len[i] - 1 - s) : s;
            const unsigned vo = on ? p_g0[i] + (unsigned)t * 16u : kOOB;
            const u32x4 v = *reinterpret_cast<const u32x4*>(hb + p_lds[i]);
            __builtin_amdgcn_raw_buffer_store_b128(v, ors, vo, 0, 0);
        }
    };
    auto store_pass_slow = [&](int s, const unsigned char* hb) {   // H % 8 != 0: 2-byte pieces
        for (int i = wave * LPW; i < wave * LPW + LPW && i < M; ++i) {
            const int len = lens_s[i];
            if (s < len) {
                const int t = rev ? (len - 1 - s) : s;
                const size_t rowi = (size_t)(n0 + i) * a.T + t;
                const __bf16* src = reinterpret_cast<const __bf16*>(hb + i * RS);
                for (int k = lane; k < a.H; k += 64) {
                    const int f = dir * a.H + k;
                    const size_t o = ((size_t)(f >> 3) * rows_total + rowi) * 8 + (f & 7);
                    a.out[o] = src[k];
                    a.out[a.out_plane + o] = *reinterpret_cast<const __bf16*>(reinterpret_cast<const unsigned char*>(src + k) + plane);
                }
            }
        }
    };
    const bool fast_out = (a.H & 7) == 0;

    // ---- prologue: weights of the first D stages, xproj of step 0
#pragma unroll
    for (int d = 0; d < D; ++d) load_w(d % Q, ring[d]);
    {
        unsigned vo[NT];
#pragma unroll
        for (int g = 0; g < NT; ++g) vo[g] = xvoff(0, g);
#pragma unroll
        for (int j = 0; j < NBW; ++j) load_x(j, vo);
    }

    for (int s = 0; s < Lmax; ++s) {
        // the ~100 scalar offsets (base + constant) are cheap to make and expensive to keep: stop the compiler from hoisting
        // them out of the time loop into (spilled) SGPRs
        asm volatile("" : "+s"(wso0), "+s"(xso0));
        const unsigned char* hcur = hs + (s & 1) * buf;          // h_{s-1}
        unsigned char* hnext = hs + ((s & 1) ^ 1) * buf;         // h_s
        // h fragments of every K block: resident for the whole step
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                const unsigned char* hp = hcur + (16 * g + line) * RS + (kb * 32 + us * 8) * 2;
                hh[kb][g] = *reinterpret_cast<const bf16x8*>(hp);
                if (kb < HLR) hl[kb][g] = *reinterpret_cast<const bf16x8*>(hp + plane);
            }
        if (s > 0) {
            if (fast_out) store_pass(s - 1, hcur);
            else store_pass_slow(s - 1, hcur);
        }
        unsigned vo_cur[NT], vo_nxt[NT];
#pragma unroll
        for (int g = 0; g < NT; ++g) { vo_cur[g] = xvoff(s, g); vo_nxt[g] = xvoff(s + 1, g); }

#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            // xproj bursts: refill what was consumed since the previous burst
            if (j == P1) {
#pragma unroll
                for (int jj = P2; jj < NBW; ++jj) load_x(jj, vo_cur);      // consumed in the previous step, needed later in this one
#pragma unroll
                for (int jj = 0; jj < P1; ++jj) load_x(jj, vo_nxt);
            }
            if (j == P2) {
#pragma unroll
                for (int jj = P1; jj < P2 && jj < NBW; ++jj) load_x(jj, vo_nxt);
            }
            // NT >= 2: two accumulators per tile (main term, cross terms); NT == 1: three, so that no MFMA waits on its
            // immediate predecessor
            constexpr int NA = NT == 1 ? 3 : 2;
            f32x4 acc[NA][NT];
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                acc[0][g] = xr[j][g];
#pragma unroll
                for (int i = 1; i < NA; ++i) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                constexpr int dummy = 0; (void)dummy;
                const int q = j * NKB + kb;
                const int slot = q % D;
                const bf16x8 whi = as_bf(ring[slot].hi), wlo = as_bf(ring[slot].lo);
                bf16x8 hlk[NT];
#pragma unroll
                for (int g = 0; g < NT; ++g) {
                    if (kb < HLR) hlk[g] = hl[kb < HLR ? kb : 0][g];
                    else hlk[g] = *reinterpret_cast<const bf16x8*>(hcur + (16 * g + line) * RS + (kb * 32 + us * 8) * 2 + plane);
                }
                if (!KRK_DBGBIT(a, 4)) {
#pragma unroll
                    for (int g = 0; g < NT; ++g) acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, hh[kb][g], acc[0][g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NT; ++g) acc[1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, hlk[g], acc[1][g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NT; ++g) acc[NA - 1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, hh[kb][g], acc[NA - 1][g], 0, 0, 0);
                }
                load_w((q + D) % Q, ring[slot]);     // wraps into the next step: the stream never drains
            }
            // ---- gates of block j: one (line, unit) per lane per tile
            if (!KRK_DBGBIT(a, 2)) {
                const int unit = (wave + NW * j) * 4 + us;
#pragma unroll
                for (int g = 0; g < NT; ++g) {
                    f32x4 z = acc[0][g] + acc[1][g];
                    if constexpr (NA == 3) z = acc[0][g] + (acc[1][g] + acc[2][g]);
                    const float gi = krk_sigmoid(z[0]);
                    const float gf = krk_sigmoid(z[1]);
                    const float gg = krk_tanh(z[2]);
                    const float go = krk_sigmoid(z[3]);
                    const float c = gf * cst[j][g] + gi * gg;
                    cst[j][g] = c;
                    const float h = go * krk_tanh(c);
                    const __bf16 hb = (__bf16)h;
                    __bf16* dst = reinterpret_cast<__bf16*>(hnext + (16 * g + line) * RS) + unit;
                    dst[0] = hb;
                    *reinterpret_cast<__bf16*>(reinterpret_cast<unsigned char*>(dst) + plane) = (__bf16)(h - (float)hb);
                }
            }
        }
        if constexpr (P2 >= NBW) {          // second burst after the last block
#pragma unroll
            for (int jj = P1; jj < NBW; ++jj) load_x(jj, vo_nxt);
        }
        // ring slot of stage q is q % D; the next step restarts at q = 0: rotate when Q is not a multiple of D
        if constexpr (Q % D != 0) {
            WFrag tmp[D];
#pragma unroll
            for (int d = 0; d < D; ++d) tmp[d] = ring[(d + Q) % D];
#pragma unroll
            for (int d = 0; d < D; ++d) ring[d] = tmp[d];
        }
        __syncthreads();
    }
    if (Lmax > 0) {
        const unsigned char* hlast = hs + (Lmax & 1) * buf;
        if (fast_out) store_pass(Lmax - 1, hlast);
        else store_pass_slow(Lmax - 1, hlast);
    }
}

template <int NW, int NT, int NKB, int NBW>
__device__ __forceinline__ void lstm_v2_wave(const LstmX3Args& a, unsigned char* hs, const int* lens_s, int Lmax,
                                             int wave, int lane, int dir, bool rev, int n0) {
    if constexpr (NBW == 0) {
        // a wave without gate columns still takes part in the step barriers and the output pass
        constexpr int M = 16 * NT;
        const int buf = 2 * M * a.hrow;
        constexpr int LPW = (M + NW - 1) / NW;
        const size_t rows_total = (size_t)a.N * a.T;
        for (int s = 0; s <= Lmax; ++s) {
            if (s > 0) {
                const unsigned char* hb = hs + (s & 1) * buf;
                for (int i = wave * LPW; i < wave * LPW + LPW && i < M; ++i) {
                    const int len = lens_s[i];
                    if (s - 1 < len) {
                        const int t = rev ? (len - s) : s - 1;
                        const size_t rowi = (size_t)(n0 + i) * a.T + t;
                        const __bf16* src = reinterpret_cast<const __bf16*>(hb + i * a.hrow);
                        for (int k = lane; k < a.H; k += 64) {
                            const int f = dir * a.H + k;
                            const size_t o = ((size_t)(f >> 3) * rows_total + rowi) * 8 + (f & 7);
                            a.out[o] = src[k];
                            a.out[a.out_plane + o] = *reinterpret_cast<const __bf16*>(reinterpret_cast<const unsigned char*>(src + k) + M * a.hrow);
                        }
                    }
                }
            }
            if (s < Lmax) __syncthreads();
        }
    } else {
        // the two waves of a SIMD (w, w + NW/2 for NW = 8) fetch xproj a quarter step apart
        constexpr int PA1 = 0, PA2 = NBW > 1 ? (NBW + 1) / 2 : 1;
        constexpr int PB1 = NBW >= 4 ? NBW / 4 : 0, PB2 = NBW >= 4 ? (3 * NBW) / 4 : PA2;
        if (NW == 8 && (wave & 4))
            lstm_v2_loop<NW, NT, NKB, NBW, PB1, PB2>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
        else
            lstm_v2_loop<NW, NT, NKB, NBW, PA1, PA2>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
    }
}

template <int NW, int NT, int NKB, int MAXB>
__global__ void __launch_bounds__(64 * NW) lstm_x3v2_kernel(const LstmX3Args a) {
    constexpr int M = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    unsigned char* hs = smem8;                                       // [2 buffers][2 planes][M][hrow]
    int* lens_s = reinterpret_cast<int*>(smem8 + 4 * M * a.hrow);   // [M]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // direction = workgroup id % ndir: workgroups are dealt to XCDs round-robin by id, so with two directions the even
    // XCDs' L2s only ever hold the forward weights and the odd ones the reverse weights
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
    if (nb_mine == MAXB) lstm_v2_wave<NW, NT, NKB, MAXB>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
    else lstm_v2_wave<NW, NT, NKB, MAXB - 1>(a, hs, lens_s, Lmax, wave, lane, dir, rev, n0);
}

template <int NW, int NT, int NKB, int MAXB>
int launch_v2(const LstmX3Args& a, hipStream_t s) {
    constexpr int M = 16 * NT;
    dim3 grid((unsigned)((a.N + M - 1) / M * a.ndir));
    const size_t lds = (size_t)4 * M * a.hrow + M * sizeof(int);
    auto kfn = lstm_x3v2_kernel<NW, NT, NKB, MAXB>;
    if (lds > 48 * 1024) {
        // per device: the attribute belongs to the function object of the CURRENT device
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(kfn, grid, dim3(64 * NW), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int NW, int NT>
int dispatch_v2(const LstmX3Args& a, hipStream_t s) {
    const int maxb = (a.NB + NW - 1) / NW;
    // NB = Hp/4 with Hp a multiple of 4 and NKB = ceil(Hp/32): for NW = 8, ceil(NB/8) == NKB; for NW = 4 it is 2*NKB-1 or 2*NKB
#define KRK_V2(NKB_, MAXB_) if (a.NKB == NKB_ && maxb == MAXB_) return launch_v2<NW, NT, NKB_, MAXB_>(a, s)
    if constexpr (NW == 8) {
        KRK_V2(1, 1); KRK_V2(2, 2); KRK_V2(3, 3); KRK_V2(4, 4); KRK_V2(5, 5); KRK_V2(6, 6); KRK_V2(7, 7); KRK_V2(8, 8);
    } else {
        KRK_V2(7, 13);
    }
#undef KRK_V2
    return -4;
}

}  // namespace

// Lines per workgroup (16 * nt) trade latency against chip time: nt = 1 holds 2*N/16 CUs and is stream bound (the whole
// W_hh per step per CU), nt = 2 halves the CUs at nearly the same step time, nt = 4 is MFMA bound.
int krk_launch_lstm_x3v2(const LstmX3Args& a, int nt, int nw, hipStream_t s) {
    if (a.NKB < 1 || a.NKB > 8 || a.NB > 64) return -4;
    if ((size_t)a.out_plane * 4 >= 0x80000000ull) return -4;                                  // 32-bit buffer offsets, kOOB
    if ((size_t)16 * nt * a.T * a.xstride * 4 >= 0x80000000ull) return -4;
    if (nw == 4) {
        if (nt == 4) return dispatch_v2<4, 4>(a, s);
        if (nt == 2) return dispatch_v2<4, 2>(a, s);
        return -4;
    }
    if (nt == 1) return dispatch_v2<8, 1>(a, s);
    if (nt == 2) return dispatch_v2<8, 2>(a, s);
    return -4;
}
