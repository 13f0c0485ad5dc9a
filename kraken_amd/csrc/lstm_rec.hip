// Recurrent part of the (bi)directional LSTM over the width axis.
//
// Replaces torch.nn.LSTM as driven by TransposedSummarizingRNN.forward
// (reference kraken/lib/vgsl/layers.py:513-547: pack_padded_sequence -> nn.LSTM ->
// pad_packed_sequence): gates i,f,g,o; c' = sig(f) c + sig(i) tanh(g); h' = sig(o) tanh(c');
// the reverse direction starts at each line's own last valid step; outputs past a
// line's length stay zero (the caller pre-zeroes `out`).
//
// Design: one workgroup = (tile of M lines, one direction) and runs ALL time steps
// with no inter-workgroup communication.  Per step it computes the M x 4H gate
// pre-activations  acc = xproj[t] + h_{t-1} . W_hh^T  on the f32 matrix cores:
//   - h_{t-1} lives in LDS as hs[k][line] (A operand: 32/16 consecutive lines per lane group);
//   - W_hh streams from L2 in B-fragment order.  The stream is the critical resource
//     (0.64 MB per step per workgroup for H=200): weights are packed so that one lane's
//     fragments for KG consecutive K-steps are contiguous (dwordx4 loads, 1-2 KB per wave
//     instruction) and the loop is software-pipelined over two register buffers -- the
//     loads of K-group g+1 are in flight while the MFMAs of group g issue;
//   - xproj[t+1] (input projection + both biases, produced by the GEMM in conv_mfma.hip)
//     is prefetched into registers during step t and initialises the accumulators;
//   - gate columns are interleaved (col = 4*unit + gate) by the weight packer so that the
//     four gates of a hidden unit sit in the four lanes of a DPP quad: the cell update
//     needs three quad broadcasts and no LDS round trip; c stays in registers for the
//     whole sequence;
//   - h_t goes to the other LDS buffer (one barrier per step) and is streamed to `out`
//     with coalesced stores while the next step's loads are in flight.
// M = 32 uses v_mfma_f32_32x32x2_f32, M = 16 uses v_mfma_f32_16x16x4_f32 (twice the
// workgroups for small batches, same FLOP rate).
#include "common.h"
#include <type_traits>

namespace {

template <int CTRL>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ f32x16 mma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// KG consecutive K-steps of one column block, one lane: KG floats = KG/4 dwordx4 loads.
template <int KG>
struct WFrag {
    f32x4 v[KG / 4];
    __device__ __forceinline__ void load(const float* p) {
#pragma unroll
        for (int i = 0; i < KG / 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + 4 * i);
    }
    __device__ __forceinline__ float at(int e) const { return v[e >> 2][e & 3]; }
};

// Per-wave state that survives the whole sequence.
template <int M>
struct LstmGeom {
    static constexpr int NACC = (M == 32) ? 16 : 4;   // accumulator registers per column block
    static constexpr int KPI = (M == 32) ? 2 : 4;     // K per MFMA
    static constexpr int LS = M + 1;                  // LDS line stride of hs (odd: conflict-free both ways)
    static constexpr int UPB = M / 4;                 // hidden units per column block
    using accv = typename std::conditional<M == 32, f32x16, f32x4>::type;
};

// The time loop for a wave that owns exactly NBW column blocks (wave, wave+4, ...).  NBW is a
// compile-time constant so that the K loop is one straight-line block of loads and MFMAs: a
// run-time block count would put a scalar branch around every MFMA and stop the scheduler from
// overlapping the weight stream with the matrix pipe.
template <int M, int NBW, int KG, bool XPRE>
__device__ __forceinline__ void lstm_time_loop(const LstmArgs& a, float* hs, const int* lens_s, int hrows, int Lmax,
                                               int wave, int lane, int dir, bool rev, int n0) {
    using G = LstmGeom<M>;
    constexpr int NACC = G::NACC, KPI = G::KPI, LS = G::LS, UPB = G::UPB;
    using accv = typename G::accv;

    const int cl = lane & (M - 1);            // column inside a block
    const int gate = cl & 3, ul = cl >> 2;
    const int khalf = (M == 32) ? (lane >> 5) : (lane >> 4);
    const int arow = lane & (M - 1);

    int irow[NACC], ilen[NACC];
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
        irow[r] = (M == 32) ? ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) : (4 * (lane >> 4) + r);
        ilen[r] = lens_s[irow[r]];
    }

    accv acc[NBW];
    accv xn[XPRE ? NBW : 1];   // next step's input projection (prefetch), if registers allow
    float cst[NBW][NACC];
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int r = 0; r < NACC; ++r) cst[j][r] = 0.f;

    // weights: [dir][group][block][lane][KG]
    const float* wbase = a.wp + ((size_t)dir * a.NG * a.NB * 64 + lane) * KG;
    const size_t gstride = (size_t)a.NB * 64 * KG;
    const float gscale = (gate == 2) ? 2.f : 1.f;   // tanh(x) = 2 sig(2x) - 1 for the cell gate

    auto load_x = [&](int s, auto& dst) {
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const bool on = s < ilen[r];
            const int t = rev ? (ilen[r] - 1 - s) : s;
            const float* xr = a.xp + ((size_t)(n0 + irow[r]) * a.T + (on ? t : 0)) * a.xstride + (size_t)dir * a.G + cl;
#pragma unroll
            for (int j = 0; j < NBW; ++j) dst[j][r] = on ? xr[(size_t)(wave + 4 * j) * M] : 0.f;
        }
    };
    auto load_w = [&](int g, WFrag<KG> (&dst)[NBW]) {
        const float* wg = wbase + (size_t)g * gstride;
#pragma unroll
        for (int j = 0; j < NBW; ++j) dst[j].load(wg + (size_t)(wave + 4 * j) * 64 * KG);
    };
    auto mma_group = [&](int g, const float* hcur, const WFrag<KG> (&w)[NBW]) {
        float av[KG];
#pragma unroll
        for (int e = 0; e < KG; ++e) av[e] = hcur[(KPI * (g * KG + e) + khalf) * LS + arow];
#pragma unroll
        for (int e = 0; e < KG; ++e)
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[j] = mma(av[e], w[j].at(e), acc[j]);
    };

    WFrag<KG> wa[NBW], wb[NBW];
    load_w(0, wa);                       // group 0 of step 0
    if constexpr (XPRE) load_x(0, xn);
    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        if constexpr (XPRE) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[j] = xn[j];
            if (s + 1 < Lmax && !KRK_DBGBIT(a, 8)) load_x(s + 1, xn);   // in flight for the whole step
        } else {
            load_x(s, acc);
        }

        // ---- acc += h_{t-1} . W_hh^T, weight stream double-buffered in registers
        const float* hcur = hs + cur * hrows * LS;
        int g = 0;
        if KRK_DBGBIT(a, 1) g = a.NG;          // probe: skip the recurrent GEMM
        for (; g + 1 < a.NG; g += 2) {
            load_w(g + 1, wb);
            mma_group(g, hcur, wa);
            load_w(g + 2 < a.NG ? g + 2 : 0, wa);     // wraps to group 0 of the NEXT step
            mma_group(g + 1, hcur, wb);
        }
        if (g < a.NG) {
            mma_group(g, hcur, wa);
            load_w(0, wa);                             // group 0 of the next step
        }

        // ---- gate non-linearities, cell update, h_t -> LDS
        float* hnext = hs + (cur ^ 1) * hrows * LS;
        if (!KRK_DBGBIT(a, 2))                 // probe: skip the gate math
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int unit = (wave + 4 * j) * UPB + ul;
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                float gv = __builtin_amdgcn_rcpf(1.0f + __expf(-gscale * acc[j][r]));
                gv = (gate == 2) ? (2.f * gv - 1.f) : gv;
                const float gi = quad_bcast<0x00>(gv);
                const float gf = quad_bcast<0x55>(gv);
                const float gg = quad_bcast<0xAA>(gv);
                const float go = quad_bcast<0xFF>(gv);
                const float c = gf * cst[j][r] + gi * gg;
                cst[j][r] = c;
                const float h = go * krk_tanh(c);
                if (gate == 0) hnext[unit * LS + irow[r]] = h;
            }
        }
        __syncthreads();
        // ---- h_t -> out[n][t][dir*H + k], coalesced
        for (int i = wave; i < M; i += 4) {
            const int li = lens_s[i];
            if (s < li) {
                const int t = rev ? (li - 1 - s) : s;
                float* o = a.out + ((size_t)(n0 + i) * a.T + t) * a.ostride + (size_t)dir * a.H;
                for (int k = lane; k < a.H; k += 64) o[k] = hnext[k * LS + i];
            }
        }
        cur ^= 1;
    }
}

// A wave without column blocks (4*Hp/M < 4) still takes part in the barriers and the output pass.
template <int M>
__device__ __forceinline__ void lstm_idle_loop(const LstmArgs& a, const float* hs, const int* lens_s, int hrows,
                                               int Lmax, int wave, int lane, int dir, bool rev, int n0) {
    constexpr int LS = LstmGeom<M>::LS;
    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        const float* hnext = hs + (cur ^ 1) * hrows * LS;
        __syncthreads();
        for (int i = wave; i < M; i += 4) {
            const int li = lens_s[i];
            if (s < li) {
                const int t = rev ? (li - 1 - s) : s;
                float* o = a.out + ((size_t)(n0 + i) * a.T + t) * a.ostride + (size_t)dir * a.H;
                for (int k = lane; k < a.H; k += 64) o[k] = hnext[k * LS + i];
            }
        }
        cur ^= 1;
    }
}

// MAXB = ceil(NB / 4): every wave owns MAXB or MAXB-1 column blocks.
template <int M, int MAXB, int KG, bool XPRE>
__global__ void __launch_bounds__(256, 1) lstm_f32_kernel(const LstmArgs a) {
    using G = LstmGeom<M>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hrows = a.NG * KG * G::KPI;              // K rows incl. zero padding of the last group
    float* hs = smem;                                  // [2][hrows][LS]
    int* lens_s = reinterpret_cast<int*>(smem + 2 * hrows * G::LS);  // [M]

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
    for (int e = tid; e < 2 * hrows * G::LS; e += 256) hs[e] = 0.f;
    __syncthreads();
    int Lmax = 0;
    for (int i = 0; i < M; ++i) Lmax = max(Lmax, lens_s[i]);

    const int nb_mine = (a.NB - wave + 3) / 4;   // blocks wave, wave+4, ...
    if (nb_mine == MAXB) {
        lstm_time_loop<M, MAXB, KG, XPRE>(a, hs, lens_s, hrows, Lmax, wave, lane, dir, rev, n0);
    } else {
        if constexpr (MAXB > 1)
            lstm_time_loop<M, MAXB - 1, KG, XPRE>(a, hs, lens_s, hrows, Lmax, wave, lane, dir, rev, n0);
        else
            lstm_idle_loop<M>(a, hs, lens_s, hrows, Lmax, wave, lane, dir, rev, n0);
    }
}

// Hidden sizes above 256 (round 4; the reference builds nn.LSTM for any width, kraken/lib/vgsl/model.py:579-593,
// layers.py:504-511): the same arithmetic as lstm_time_loop on 16-line tiles (v_mfma_f32_16x16x4_f32, weights in the M = 16 /
// KG = 4 fragment order), but the gate-column blocks of a wave are a RUN-TIME loop -- each block runs its whole K loop, then its
// gate math -- and the cell state lives in LDS, so nothing is sized by the hidden width except LDS (h twice + c: 12 bytes x 17 per
// unit: Hp <= 768).  A correctness path: W_hh streams from L2 every step (4 Hp^2 floats per workgroup), no prefetch pipelining.
// Round 6 (the reference accepts any size): above 768 the CELL STATE lives in HBM (CST: only the gate-0 lane of a
// unit's quad reads and writes it -- a thread sees its own stores -- and broadcasts it where the peephole cell needs it; LDS then
// holds h twice: Hp <= 1152); above that h lives in HBM as well (HST: [parity][K row][16 lines] per workgroup, written with plain
// stores -- write-through to L2, completed by the barrier's vmcnt(0) -- and read with agent-scope loads that bypass the CU's L1):
// no LDS limit at all.  The arithmetic and its order are those of the LDS form (same MFMA sequence per block).
template <bool CST, bool HST>
__global__ void __launch_bounds__(256, 1) lstm_big_kernel(const LstmArgs a) {
    constexpr int M = 16, KG = 4, KPI = 4;
    constexpr int LS = HST ? M : M + 1;                // row stride of h (LDS rows are padded to 17 banks; HBM rows are not)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hrows = a.NG * KG * KPI;                 // K rows incl. the zero padding of the last group
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    float* hs = HST ? a.hstate + wg * 2 * hrows * LS : smem;                       // [2][hrows][LS]
    float* cs = CST ? a.cstate + wg * (size_t)a.Hp * M : smem + 2 * hrows * LS;    // [Hp][CLS] cell state of (unit, line)
    constexpr int CLS = CST ? M : M + 1;
    int* lens_s = reinterpret_cast<int*>(smem + (HST ? 0 : 2 * hrows * LS) + (CST ? 0 : a.Hp * CLS));

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
    for (int e = tid; e < (HST ? 0 : 2 * hrows * LS) + (CST ? 0 : a.Hp * CLS); e += 256) smem[e] = 0.f;
    __syncthreads();
    int Lmax = 0;
    for (int i = 0; i < M; ++i) Lmax = max(Lmax, lens_s[i]);

    const int cl = lane & 15, gate = cl & 3, ul = cl >> 2, khalf = lane >> 4, arow = lane & 15;
    int irow[4], ilen[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        irow[r] = 4 * (lane >> 4) + r;
        ilen[r] = lens_s[irow[r]];
    }
    const float* wbase = a.wp + ((size_t)dir * a.NG * a.NB * 64 + lane) * KG;
    const size_t gstride = (size_t)a.NB * 64 * KG;
    const float gscale = (gate == 2) ? 2.f : 1.f;
    auto h_at = [&](const float* hb, int idx) -> float {       // h of the previous step: LDS, or HBM past this CU's L1
        if constexpr (HST) return __hip_atomic_load(hb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return hb[idx];
    };
    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        const float* hcur = hs + cur * hrows * LS;
        float* hnext = hs + (cur ^ 1) * hrows * LS;
        // FOUR gate-column blocks of a wave advance together (round 6): four independent accumulator chains share every h operand
        // (one wave per SIMD: a single chain of dependent 16x16x4 MFMAs left the matrix pipe idle between them), and the weight
        // fragments of four K groups x four blocks are requested before the first is used (one load at a time, each waited for in
        // front of its MFMAs, made a step of H = 512 a chain of 1024 L2 latencies: 53 ms per layer).  Per accumulator the K
        // order is unchanged.
        constexpr int NA = 4, PF = 4;
        for (int b0 = wave; b0 < a.NB; b0 += 4 * NA) {
            f32x4 acc[NA];
            const float* wb[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int b = min(b0 + 4 * j, a.NB - 1);            // (blocks past the end recompute the last one; never stored)
                wb[j] = wbase + (size_t)b * 64 * KG;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool on = s < ilen[r];
                    const int t = rev ? (ilen[r] - 1 - s) : s;
                    const float* xr = a.xp + ((size_t)(n0 + irow[r]) * a.T + (on ? t : 0)) * a.xstride + (size_t)dir * a.G + cl;
                    acc[j][r] = on ? xr[(size_t)b * M] : 0.f;
                }
            }
            int g = 0;
            for (; g + PF <= a.NG; g += PF) {
                f32x4 w[NA][PF];
#pragma unroll
                for (int u = 0; u < PF; ++u)
#pragma unroll
                    for (int j = 0; j < NA; ++j) w[j][u] = *reinterpret_cast<const f32x4*>(wb[j] + (size_t)(g + u) * gstride);
#pragma unroll
                for (int u = 0; u < PF; ++u)
#pragma unroll
                    for (int e = 0; e < KG; ++e) {
                        const float hv = h_at(hcur, (KPI * ((g + u) * KG + e) + khalf) * LS + arow);
#pragma unroll
                        for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv, w[j][u][e], acc[j], 0, 0, 0);
                    }
            }
            for (; g < a.NG; ++g) {
                f32x4 w[NA];
#pragma unroll
                for (int j = 0; j < NA; ++j) w[j] = *reinterpret_cast<const f32x4*>(wb[j] + (size_t)g * gstride);
#pragma unroll
                for (int e = 0; e < KG; ++e) {
                    const float hv = h_at(hcur, (KPI * (g * KG + e) + khalf) * LS + arow);
#pragma unroll
                    for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv, w[j][e], acc[j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < NA; ++j) {
            const int b = b0 + 4 * j;
            if (b >= a.NB) break;
            const f32x4 accb = acc[j];
            const int unit = b * 4 + ul;
            // the cell state in front of this step: LDS (every lane of the quad reads it), or HBM through the quad's gate-0 lane
            auto c_prev = [&](int r) -> float {
                if constexpr (CST) return quad_bcast<0x00>((gate == 0 && s > 0) ? cs[unit * CLS + irow[r]] : 0.f);
                else return cs[unit * CLS + irow[r]];
            };
            if (a.peep) {
                // ocropy's peephole cell (reference layers.py:72-103): i and f look at c, the output gate at the NEW c and is not
                // squashed: c' = sig(f + w_f c) c + sig(i + w_i c) tanh(g); h = (o + w_o c') tanh(c')
                const float* pw = a.peep + (size_t)dir * 3 * a.Hp;
                const float wp = gate < 2 ? pw[gate * a.Hp + unit] : 0.f, wo = pw[2 * a.Hp + unit];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float cp = c_prev(r);
                    const float z = accb[r] + wp * cp;
                    float gv = __builtin_amdgcn_rcpf(1.0f + __expf(-gscale * z));
                    gv = (gate == 2) ? (2.f * gv - 1.f) : (gate == 3 ? z : gv);
                    const float gi = quad_bcast<0x00>(gv);
                    const float gf = quad_bcast<0x55>(gv);
                    const float gg = quad_bcast<0xAA>(gv);
                    const float zo = quad_bcast<0xFF>(gv);
                    const float c = gf * cp + gi * gg;
                    const float h = (zo + wo * c) * krk_tanh(c);
                    if (gate == 0) {
                        cs[unit * CLS + irow[r]] = c;
                        hnext[unit * LS + irow[r]] = h;
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gv = __builtin_amdgcn_rcpf(1.0f + __expf(-gscale * accb[r]));
                gv = (gate == 2) ? (2.f * gv - 1.f) : gv;
                const float gi = quad_bcast<0x00>(gv);
                const float gf = quad_bcast<0x55>(gv);
                const float gg = quad_bcast<0xAA>(gv);
                const float go = quad_bcast<0xFF>(gv);
                const float c = gf * c_prev(r) + gi * gg;
                const float h = go * krk_tanh(c);
                if (gate == 0) {
                    cs[unit * CLS + irow[r]] = c;
                    hnext[unit * LS + irow[r]] = h;
                }
            }
            }
        }
        __syncthreads();                // (HST: s_waitcnt vmcnt(0) in front of the barrier -- every wave's h stores have reached L2)
        for (int i = wave; i < M; i += 4) {
            const int li = lens_s[i];
            if (s < li) {
                const int t = rev ? (li - 1 - s) : s;
                float* o = a.out + ((size_t)(n0 + i) * a.T + t) * a.ostride + (size_t)dir * a.H;
                for (int k = lane; k < a.H; k += 64) o[k] = h_at(hnext, k * LS + i);
            }
        }
        cur ^= 1;
    }
}

template <int M, int MAXB, int KG, bool XPRE>
int launch_one(const LstmArgs& a, hipStream_t s) {
    dim3 grid((unsigned)((a.N + M - 1) / M), (unsigned)a.ndir);
    constexpr int KPI = (M == 32) ? 2 : 4;
    const size_t lds = ((size_t)2 * a.NG * KG * KPI * (M + 1) + M) * sizeof(float);
    auto kfn = lstm_f32_kernel<M, MAXB, KG, XPRE>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// K-steps per lane-contiguous weight group for a given tile size / blocks-per-wave; the packer
// (capi.hip) asks for the same value.  Chosen so that accumulators + cell state + prefetch
// buffers stay inside the 512-register budget of one wave per SIMD.
int krk_lstm_kg(int M, int per_wave) {
    if (M == 32) return 4;
    return per_wave <= 13 ? 8 : 4;
}

// LDS holds h twice and the cell state up to Hp = 768; the cell state moves to HBM above that, h above 1152 (see lstm_big_kernel)
constexpr int kBigCellInLds = 768, kBigHInLds = 1152;
size_t krk_lstm_big_cstate_floats(int N, int ndir, int Hp) {
    return Hp > kBigCellInLds ? (size_t)((N + 15) / 16) * ndir * Hp * 16 : 0;
}
size_t krk_lstm_big_hstate_floats(int N, int ndir, int Hp) {
    const int hrows = (Hp / 4 + 3) / 4 * 16;
    return Hp > kBigHInLds ? (size_t)((N + 15) / 16) * ndir * 2 * hrows * 16 : 0;
}

// hidden sizes Hp > 256 (or the peephole cell at any width): a.NB = Hp / 4 blocks of 16 gate columns, a.NG = K groups of 4 steps
// (weights: the M = 16 pack).  The caller provides a.cstate / a.hstate (sizes above; hstate ZEROED) when they are non-zero.
int krk_launch_lstm_big(const LstmArgs& a, hipStream_t s) {
    if (a.N <= 0) return 0;
    const bool cst = a.Hp > kBigCellInLds, hst = a.Hp > kBigHInLds;
    if ((cst && !a.cstate) || (hst && !a.hstate)) return -4;
    dim3 grid((unsigned)((a.N + 15) / 16), (unsigned)a.ndir);
    const size_t lds = ((hst ? 0 : (size_t)2 * a.NG * 16 * 17) + (cst ? 0 : (size_t)a.Hp * 17) + 16) * sizeof(float);
    auto launch = [&](auto kfn) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);
    };
    if (hst) launch(lstm_big_kernel<true, true>);
    else if (cst) launch(lstm_big_kernel<true, false>);
    else launch(lstm_big_kernel<false, false>);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// M in {16, 32}; a.NB = 4*Hp / M column blocks; a.NG = K-groups of krk_lstm_kg() steps.
int krk_launch_lstm(const LstmArgs& a, int M, hipStream_t s) {
    const int per_wave = (a.NB + 3) / 4;
#define KRK_CASE(M_, B_, KG_, X_) case B_: return launch_one<M_, B_, KG_, X_>(a, s)
    if (M == 32) {
        switch (per_wave) {
            KRK_CASE(32, 1, 4, true); KRK_CASE(32, 2, 4, true); KRK_CASE(32, 3, 4, true); KRK_CASE(32, 4, 4, true);
            KRK_CASE(32, 5, 4, true); KRK_CASE(32, 6, 4, false); KRK_CASE(32, 7, 4, false); KRK_CASE(32, 8, 4, false);
            default: return -4;
        }
    }
    switch (per_wave) {
        KRK_CASE(16, 1, 8, true); KRK_CASE(16, 2, 8, true); KRK_CASE(16, 3, 8, true); KRK_CASE(16, 4, 8, true);
        KRK_CASE(16, 5, 8, true); KRK_CASE(16, 6, 8, true); KRK_CASE(16, 7, 8, true); KRK_CASE(16, 8, 8, true);
        KRK_CASE(16, 9, 8, true); KRK_CASE(16, 10, 8, true); KRK_CASE(16, 11, 8, true); KRK_CASE(16, 12, 8, true);
        KRK_CASE(16, 13, 8, true); KRK_CASE(16, 14, 4, true); KRK_CASE(16, 15, 4, true); KRK_CASE(16, 16, 4, true);
        default: return -4;
    }
#undef KRK_CASE
}
