// Wide-tile row projection  Y[M][Cout] = X[M][K] . W^T + b  with split-bf16 operands (round 4): the LSTM input projection
// of every time step of every line (torch.nn.LSTM's W_ih x_t + b_ih + b_hh, reference kraken/lib/vgsl/layers.py:507-511) on
// large batches.  Same operand layouts and the same results as gemm_x3.hip (which stays the kernel for narrow outputs and
// small row counts); what changes is the shape of the work on a CU:
//
//   tile      = 256 rows x 64*NCB columns per workgroup of EIGHT waves (4 along rows x 2 along columns; a wave owns 64 rows x
//               32*NCB columns = 2*NCB accumulators of 32x32).  NCB = 5 covers Cout = 1600 (two directions x four gates x 200
//               units) in exactly five column groups; the operand stream L2 -> LDS per MFMA is 2731 x (1/256 + 1/320) = 19 B/clk
//               per CU at full matrix rate, where gemm_x3's 256 x 128 tile needs 32 (1.2 GB per launch on the headline batch)
//   K step    = 16, NBUF = 3 | 4 LDS buffers (X 16 KB + W 4*TN*16 B each) filled by global_load_lds_dwordx4 NBUF - 1 steps ahead
//   ping-pong = the two waves of a SIMD (wave w and w + 4: column half 0 / 1) run HALF A STEP apart: in every slot (one raw
//               s_barrier each) one of them issues the step's 4 + 2*NCB ds_read_b128 while the other issues its 6*NCB MFMAs, so
//               the matrix pipe of a SIMD sees a single MFMA stream with the partner's LDS latency beside it, not in front of it
//               (gemm_x3: both workgroups of a CU read, then both multiply: 0.40 MFMA-busy).  A buffer is read in two consecutive
//               slots (first by the column-half-0 waves, then by the others) and refilled two slots later
//   epilogue  = + bias, transposed through the (now free) LDS buffers in chunks of 64 columns: every store instruction writes
//               four 256-byte row runs; optional tile-time-major row permutation exactly as in gemm_x3.hip
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int WTM = 256;         // rows per workgroup
constexpr int WA_Q = 4 * WTM;    // 16-byte pieces of the X tile per K step

#if defined(KRK_ABLATE) && !defined(KRK_BF16_ONE)
// phase cycles summed over waves: 0 prologue, 1 copy waits (vmcnt), 2 barriers, 3 copy issue, 4 fragment reads (+ lgkmcnt), 5 MFMAs, 6 epilogue; [7] = waves
__device__ unsigned long long g_x3w_phases[8];
#endif

template <int NCB, int NBUF>
__global__ void __launch_bounds__(512, 2) gemm_x3w_kernel(const GemmX3Args a) {
    constexpr int TN = 64 * NCB;             // columns per workgroup
    constexpr int B_Q = 4 * TN;              // 16-byte pieces of the W tile per K step
    constexpr int STAGE = WA_Q + B_Q;        // pieces per buffer
    constexpr int NI = STAGE / 64;           // 1 KB copy instructions per step (over the 8 waves)
    constexpr int NL = (NI + 7) / 8;         // ... per wave, at most
    static_assert(STAGE % 64 == 0, "whole copy instructions");
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];   // NBUF x STAGE pieces

    KRK_PHASES(7);
    KRK_PH_START(a);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, px = lane & 31;
    const int wm = wave & 3;                 // row quarter of the tile
    const int grp = wave >> 2;               // column half; also the ping-pong group

    // workgroup id -> (row tile, column group): id % 8 is the XCD the hardware dispatches to; the column groups of a row tile
    // run back to back on one XCD (X is read from HBM once, re-read from that XCD's L2)
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int cg = jj % a.ncg;
    const int tile = (jj / a.ncg) * 8 + xcd;
    if (tile >= a.ntiles) return;
    const int row0 = tile * WTM;
    const int nkb = a.K >> 4;
    if (a.stagger > 0) {                     // probe: spread the workgroups' epilogue store bursts over time
        const long long t0 = (long long)__builtin_readcyclecounter();
        const long long d = (long long)((id >> 3) & 3) * a.stagger;
        while ((long long)__builtin_readcyclecounter() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }

    // ---- this wave's copy instructions of a step: instruction i = wave + 8k covers pieces [64 i, 64 i + 64) of the buffer
    //      (i < 16: X pieces (plane, k-half) = i / 4, rows 64 (i % 4) + lane; else W pieces)
    const __bf16* src[NL];
    long stride[NL];
    const int nl = __builtin_amdgcn_readfirstlane((NI - wave + 7) / 8);     // NL or NL - 1
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = wave + 8 * k;
        if (i < 16) {
            const int q = i >> 2, p = q >> 1, h = q & 1;
            const int row = min(row0 + (i & 3) * 64 + lane, a.M - 1);       // rows past the end re-read the last row; never stored
            src[k] = a.x + (size_t)p * a.x_plane + ((size_t)h * a.M + row) * 8;
            stride[k] = (long)2 * a.M * 8;
        } else {
            src[k] = a.w + ((size_t)cg * nkb * B_Q + (size_t)(i - 16) * 64 + lane) * 8;
            stride[k] = (long)B_Q * 8;
        }
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](int kb, int buf) {
        f32x4* dst = lds + buf * STAGE + wave * 64;
#pragma unroll
        for (int k = 0; k < NL; ++k)
            if (k < NL - 1 || nl == NL)
                __builtin_amdgcn_global_load_lds((const void*)(src[k] + (long)kb * stride[k]), (lds_ptr)(dst + k * 512), 16, 0, 0);
    };
    // my copies of every step but the `younger` youngest have landed (younger <= NBUF - 2 steps of nl copies each)
    auto wait_older = [&](int younger) {
        switch (younger * nl) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // never: NL in {4, 5}, younger <= 2
        }
    };

    f32x16 acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][s][r] = 0.f;

    const bool no_mma = KRK_DBGBIT(a, 1), no_copy = KRK_DBGBIT(a, 2), no_lds = KRK_DBGBIT(a, 8);
    const int arow = wm * 64 + px;           // first segment's row inside the tile
    const int wcol = grp * (32 * NCB) + px;  // first column block's column inside the tile
    bf16x8 xh[2], xl[2], wh[NCB], wl[NCB];
#pragma unroll
    for (int s = 0; s < 2; ++s) xh[s] = xl[s] = bf16x8{};
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) wh[cb] = wl[cb] = bf16x8{};

    auto read = [&](int buf) {
        if (no_lds) return;
        const f32x4* L = lds + buf * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            xh[s] = __builtin_bit_cast(bf16x8, L[(0 + half) * WTM + arow + 32 * s]);
            xl[s] = __builtin_bit_cast(bf16x8, L[(2 + half) * WTM + arow + 32 * s]);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            wh[cb] = __builtin_bit_cast(bf16x8, L[WA_Q + (0 + half) * TN + wcol + cb * 32]);
            wl[cb] = __builtin_bit_cast(bf16x8, L[WA_Q + (2 + half) * TN + wcol + cb * 32]);
        }
    };
    // term-major order: 2*NCB independent accumulators between two MFMAs on the same one (a wave multiplies alone on its SIMD)
    auto mma = [&]() {
        if (no_mma) return;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xh[s], acc[cb][s], 0, 0, 0);
#ifndef KRK_BF16_ONE
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[cb], xl[s], acc[cb][s], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[cb], xh[s], acc[cb][s], 0, 0, 0);
#endif
    };

    // Slots 0 .. 2 nkb, one barrier each (2 nkb + 1 for every wave).  Group 0 reads step j in slot 2j and multiplies in slot 2j + 1;
    // group 1 reads step j in slot 2j + 1 and multiplies in slot 2j + 2.  Step j's buffer (j % NBUF) is therefore read in slots 2j
    // and 2j + 1 and refilled (with step j + NBUF) in slots 2j + 2 (group 0's waves) and 2j + 3 (group 1's); all of a step's
    // copies are waited for in front of the barrier of the even slot that reads it first.
    if (!no_copy) {
#pragma unroll
        for (int k = 0; k < NBUF - 1; ++k)
            if (k < nkb) issue(k, k);
    }
    KRK_PH(a, 0);
    // Two straight-line loops, one per group (a single loop with the group as a run-time condition makes the compiler carry the
    // accumulators through phi copies and spill them).
    // a slot boundary: the MFMAs are register-only, so without the scheduling fences the compiler moves them across the barrier
    // (it sank 29 of a slot's 30 behind the NEXT barrier: both waves of a SIMD multiplying in the same slot)
    auto slot_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        KRK_PH(a, 2);
    };
    // the reads are complete before this wave arrives at the next barrier: the buffer may be refilled behind it.  Through the
    // builtin (vmcnt 63, expcnt 7, lgkmcnt 0) so that the compiler's own wait-count pass knows the fragments arrived
    auto reads_done = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); KRK_PH(a, 4); };
    // A group issues ITS waves' copies of step j + NBUF - 1 at the start of its own READ slot of step j (behind the barrier that
    // follows the last read of the buffer's previous tenant, step j - 1) -- never in front of its MFMAs: a copy instruction costs
    // the issuing wave ~20 cycles of the CU's address path per KB, and with all eight waves issuing in one slot the multiplying
    // group stood ~800 cycles behind its partner's copies (phase accounting, profiles/r04_phase_stats.txt).  A group waits for
    // its own copies of the step the NEXT even slot reads, in front of the barrier that opens it.
    auto refill = [&](int j) {
        if (j + NBUF - 1 < nkb && !no_copy) issue(j + NBUF - 1, (j + NBUF - 1) % NBUF);
        KRK_PH(a, 3);
    };
    if (grp == 0) {
        for (int j = 0; j < nkb; ++j) {
            wait_older(min(NBUF - 2, nkb - 1 - j));      // my copies of step j
            KRK_PH(a, 1);
            slot_barrier();                              // slot 2j
            refill(j);
            read(j % NBUF);
            reads_done();
            slot_barrier();                              // slot 2j + 1
            mma();
            KRK_PH(a, 5);
        }
        slot_barrier();                                  // slot 2 nkb: the other group's last MFMAs
    } else {
        wait_older(min(NBUF - 2, nkb - 1));              // my copies of step 0
        KRK_PH(a, 1);
        slot_barrier();                                  // slot 0: nothing to do yet
        for (int j = 0; j < nkb; ++j) {
            slot_barrier();                              // slot 2j + 1
            refill(j);
            read(j % NBUF);
            reads_done();
            if (j + 1 < nkb) wait_older(min(NBUF - 2, nkb - 2 - j));   // my copies of step j + 1
            KRK_PH(a, 1);
            slot_barrier();                              // slot 2j + 2
            mma();
            KRK_PH(a, 5);
        }
    }

    // ---- epilogue: D[column][row] (weights are the MFMA's A operand): lane = row px of its segment, registers 4j..4j+3 = columns
    // 8j + 4*half + 0..3 of the block.  Chunks of two column blocks are transposed through LDS (the pipeline buffers are free: the
    // barrier of the last slot is behind every read) so that every global store covers four 256-byte row runs.
    constexpr int RSTR = 64 + 4;                        // floats per LDS row: +16 B keeps the column writes conflict-free
    float* T = reinterpret_cast<float*>(lds) + wave * (32 * RSTR);
    const int colw = cg * TN + grp * (32 * NCB);        // first column of this wave
    const bool vec = (a.Cout & 3) == 0;
    const bool nostore = KRK_DBGBIT(a, 4);
    const int T16 = 16 * (a.tileT < 0 ? -a.tileT : 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int rbase = row0 + wm * 64 + 32 * s;
#pragma unroll
        for (int c0 = 0; c0 < NCB; c0 += 2) {
            const int nb = (c0 + 1 < NCB) ? 2 : 1;      // column blocks of this chunk (compile-time after unrolling)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (c0 + cb >= NCB) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[c0 + cb][s][4 * j + i];
                    *reinterpret_cast<f32x4*>(T + px * RSTR + cb * 32 + 8 * j + 4 * half) = v;
                }
            }
            // read back: nb == 2: 16 lanes per row (64 columns), 4 rows per instruction; nb == 1: 8 lanes per row, 8 rows
            const int lpr = nb == 2 ? 16 : 8;
            const int rpi = 64 / lpr;
            const int r0 = lane / lpr, cp = lane % lpr;
            const int col = colw + c0 * 32 + 4 * cp;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (vec) { if (col < a.Cout) bv = *reinterpret_cast<const f32x4*>(a.bias + col); }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e < a.Cout) bv[e] = a.bias[col + e];
            }
            int ln = 0, tt = 0, tl = 0, rem = 0;
            if (a.tileT > 0) {
                ln = (rbase + r0) / a.tileT;
                tt = (rbase + r0) - ln * a.tileT;
            } else if (a.tileT < 0) {
                tl = (rbase + r0) / T16;
                rem = (rbase + r0) - tl * T16;
            }
            for (int i = 0; i < 32 / rpi; ++i) {
                const int r = i * rpi + r0;
                f32x4 v = *reinterpret_cast<const f32x4*>(T + r * RSTR + 4 * cp);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bv[e];
                const int row = rbase + r;
                size_t orow = (size_t)row;
                bool keep = true;
                if (a.tileT > 0) {
                    orow = ((size_t)(ln >> 4) * a.tileT + tt) * 16 + (ln & 15);
                    tt += rpi;
                    while (tt >= a.tileT) { tt -= a.tileT; ++ln; }
                } else if (a.tileT < 0) {
                    const int n = tl * 16 + (rem & 15);
                    keep = n < a.nlines;
                    orow = (size_t)n * (size_t)(-a.tileT) + (rem >> 4);
                    rem += rpi;
                    while (rem >= T16) { rem -= T16; ++tl; }
                }
                if (row < a.M && keep && !nostore) {
                    float* yp = a.y + orow * a.Cout + col;
                    if (vec) {
                        if (col < a.Cout) *reinterpret_cast<f32x4*>(yp) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e < a.Cout) yp[e] = v[e];
                    }
                }
            }
        }
    }
#if defined(KRK_ABLATE) && !defined(KRK_BF16_ONE)
    KRK_PH(a, 6);
    KRK_PH_FLUSH(a, g_x3w_phases, 7);
#endif
}

template <int NCB, int NBUF>
int launch_w(const GemmX3Args& a, hipStream_t s) {
    constexpr int TN = 64 * NCB;
    const int slots = (a.ntiles + 7) / 8 * 8;
    const size_t lds = (size_t)NBUF * (WA_Q + 4 * TN) * 16;
    // the attribute belongs to the function object of the CURRENT device: once per device, not once per process
    static bool attr_set[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3w_kernel<NCB, NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_x3w_kernel<NCB, NBUF>), dim3((unsigned)(slots * a.ncg)), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

#if defined(KRK_ABLATE) && !defined(KRK_BF16_ONE)
int krk_phase_stats_x3w(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x3w_phases), sizeof(g_x3w_phases)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_x3w_phases), z, sizeof(z)) != hipSuccess) return -1; }
    return 8;
}
#endif

#ifndef KRK_BF16_ONE
// columns per workgroup of the wide kernel for `cout` output columns: the width (256 or 320) with the fewer padded columns
int krk_gemm_x3w_tn(int cout) {
    const int p256 = (cout + 255) / 256 * 256, p320 = (cout + 319) / 320 * 320;
    return p320 < p256 ? 320 : 256;
}
#endif

// a.w = [column group of tn][K/16][plane][k-half][tn columns][8]; a.ncg = column groups of tn
int KRK_FN(krk_launch_gemm_x3w)(const GemmX3Args& a, int tn, hipStream_t s) {
    if (a.K % 16 || a.M <= 0) return a.M == 0 ? 0 : -1;
    if (tn == 320) return a.nbuf == 4 ? launch_w<5, 4>(a, s) : launch_w<5, 3>(a, s);
    if (tn == 256) return a.nbuf == 4 ? launch_w<4, 4>(a, s) : launch_w<4, 3>(a, s);
    return -1;
}
