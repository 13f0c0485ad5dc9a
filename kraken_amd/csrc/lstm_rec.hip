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
//   - W_hh streams from L2 in B-fragment order (one coalesced 256-B load per fragment,
//     0.64 MB per step per CU for H=200 -- the weights of both directions stay L2-resident);
//   - xproj[t] (input projection + both biases, produced by the GEMM in conv_mfma.hip)
//     initialises the accumulators;
//   - gate columns are interleaved (col = 4*unit + gate) by the weight packer so that the
//     four gates of a hidden unit sit in the four lanes of a DPP quad: the cell update
//     needs three quad broadcasts and no LDS round trip; c stays in registers for the
//     whole sequence;
//   - h_t goes to the other LDS buffer (one barrier per step) and is streamed to `out`
//     with coalesced stores while the next step's MFMAs run.
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

template <int M, int MAXB>
__global__ void __launch_bounds__(256, 1) lstm_f32_kernel(const LstmArgs a) {
    constexpr int NACC = (M == 32) ? 16 : 4;   // accumulator registers per column block
    constexpr int KPI = (M == 32) ? 2 : 4;     // K per MFMA
    constexpr int LS = M + 1;                  // LDS line stride of hs (odd: conflict-free both ways)
    constexpr int UPB = M / 4;                 // hidden units per column block
    using accv = typename std::conditional<M == 32, f32x16, f32x4>::type;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hs = smem;                                   // [2][Hp][LS]
    int* lens_s = reinterpret_cast<int*>(smem + 2 * a.Hp * LS);  // [M]

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
    for (int e = tid; e < 2 * a.Hp * LS; e += 256) hs[e] = 0.f;
    __syncthreads();
    int Lmax = 0;
    for (int i = 0; i < M; ++i) Lmax = max(Lmax, lens_s[i]);

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
    const int nb_mine = (a.NB - wave + 3) / 4;   // blocks wave, wave+4, ...

    accv acc[MAXB];
    float cst[MAXB][NACC];
#pragma unroll
    for (int j = 0; j < MAXB; ++j)
#pragma unroll
        for (int r = 0; r < NACC; ++r) cst[j][r] = 0.f;

    const float* wbase = a.wp + (size_t)dir * a.KS * a.NB * 64 + lane;
    const float gscale = (gate == 2) ? 2.f : 1.f;   // tanh(x) = 2 sig(2x) - 1 for the cell gate

    int cur = 0;
    for (int s = 0; s < Lmax; ++s) {
        // ---- accumulators <- input projection of this step (per line: own time index)
        size_t xoff[NACC];
        bool on[NACC];
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            on[r] = s < ilen[r];
            const int t = rev ? (ilen[r] - 1 - s) : s;
            xoff[r] = ((size_t)(n0 + irow[r]) * a.T + (on[r] ? t : 0)) * a.xstride + (size_t)dir * a.G + cl;
        }
#pragma unroll
        for (int j = 0; j < MAXB; ++j) {
            if (j < nb_mine) {
                const int b = wave + 4 * j;
#pragma unroll
                for (int r = 0; r < NACC; ++r) acc[j][r] = on[r] ? a.xp[xoff[r] + (size_t)b * M] : 0.f;
            }
        }
        // ---- acc += h_{t-1} . W_hh^T
        const float* hcur = hs + cur * a.Hp * LS;
#pragma unroll 2
        for (int ks = 0; ks < a.KS; ++ks) {
            const float av = hcur[(KPI * ks + khalf) * LS + arow];
            const float* wk = wbase + (size_t)ks * a.NB * 64;
#pragma unroll
            for (int j = 0; j < MAXB; ++j) {
                if (j < nb_mine) {
                    const float wv = wk[(wave + 4 * j) * 64];
                    acc[j] = mma(av, wv, acc[j]);
                }
            }
        }
        // ---- gate non-linearities, cell update, h_t -> LDS
        float* hnext = hs + (cur ^ 1) * a.Hp * LS;
#pragma unroll
        for (int j = 0; j < MAXB; ++j) {
            if (j < nb_mine) {
                const int unit = (wave + 4 * j) * UPB + ul;
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    float g = __builtin_amdgcn_rcpf(1.0f + __expf(-gscale * acc[j][r]));
                    g = (gate == 2) ? (2.f * g - 1.f) : g;
                    const float gi = quad_bcast<0x00>(g);
                    const float gf = quad_bcast<0x55>(g);
                    const float gg = quad_bcast<0xAA>(g);
                    const float go = quad_bcast<0xFF>(g);
                    const float c = gf * cst[j][r] + gi * gg;
                    cst[j][r] = c;
                    const float h = go * krk_tanh(c);
                    if (gate == 0) hnext[unit * LS + irow[r]] = h;
                }
            }
        }
        __syncthreads();
        // ---- h_t -> out[n][t][dir*H + k], coalesced; overlaps the next step's MFMAs
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

template <int M, int MAXB>
int launch_one(const LstmArgs& a, hipStream_t s) {
    dim3 grid((unsigned)((a.N + M - 1) / M), (unsigned)a.ndir);
    const size_t lds = ((size_t)2 * a.Hp * (M + 1) + M) * sizeof(float);
    auto kfn = lstm_f32_kernel<M, MAXB>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// M in {16, 32}; a.NB = 4*Hp / M column blocks, at most 4*MAXB of them.
int krk_launch_lstm(const LstmArgs& a, int M, hipStream_t s) {
    const int per_wave = (a.NB + 3) / 4;
    if (M == 32) {
        if (per_wave <= 4) return launch_one<32, 4>(a, s);
        if (per_wave <= 8) return launch_one<32, 8>(a, s);
        return -4;
    }
    if (per_wave <= 8) return launch_one<16, 8>(a, s);
    if (per_wave <= 16) return launch_one<16, 16>(a, s);
    return -4;
}
