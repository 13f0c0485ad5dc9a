// Row-major projection  Y[M][Cout] = X[M][K] . W^T + b  on the gfx950 bf16 matrix cores with SPLIT
// operands ("bf16x3", see conv_x3.hip): the LSTM input projection of every time step of every line
// (torch.nn.LSTM's W_ih x_t + b_ih + b_hh, reference kraken/lib/vgsl/layers.py:507-511) and the
// LinSoftmax projection (layers.py:710-722).  M = lines x time steps (38 400 rows on the headline
// batch), K = 384/400, Cout = 1600 (two directions x four gates x 200) or the alphabet size.
//
// Both operands travel through LDS so that every byte is fetched from L2/HBM once per workgroup
// (four waves share one copy; per-wave weight loads saturated the 64 B/clk texture path before):
//   tile      = 256 rows x 128 columns per workgroup, 4 waves x 2 row segments x 4 column blocks
//   K step    = 16: 16 KB of X (256 rows x 16 x hi/lo) + 8 KB of W per step, THREE LDS buffers filled
//               by asynchronous global -> LDS copies (global_load_lds_dwordx4) issued two steps ahead;
//               counted vmcnt + ONE raw s_barrier per step, no staging registers, no ds_write pass
//   X layout  = K-blocked split planes [K/8][M][8 bf16] (written that way by the producers): the 64
//               lanes of one copy read 1 KB contiguous (row-major X costs one cache line per lane)
//   LDS order = [plane][k-half][row][8 bf16]: a wave's ds_read_b128 covers contiguous 512-byte runs
//   grid      = 1-D, XCD-aware: the column groups of one row tile run back-to-back on the SAME XCD,
//               so X is read from HBM once and re-read from that XCD's L2
//   row order = optionally "tile-time-major" (a.tileT = T): input row n*T + t is stored as output row
//               ((n/16)*T + t)*16 + n%16, so the 16 lines a recurrent workgroup advances together find the
//               projections of one time step in ONE contiguous 16-row run, and consecutive steps back to back:
//               the recurrent kernels' per-step xproj reads walk memory linearly instead of touching 16 rows that
//               lie T rows (~1 MB) apart -- measured 1.2 us per step of exposed xproj latency (page walks) before
//   epilogue  = + bias, transposed through the (now free) LDS buffers so that each store instruction writes
//               two full 512-byte row runs of the fp32 rows [row][Cout] the recurrent kernel / decoder read
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TM = 256;        // rows per workgroup
constexpr int TN = 128;        // columns per workgroup (4 blocks of 32)
constexpr int A_Q = 4 * TM;    // 16-byte pieces of the X tile per K step
constexpr int B_Q = 4 * TN;    // 16-byte pieces of the W tile per K step

__global__ void __launch_bounds__(256, 2) gemm_x3_kernel(const GemmX3Args a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];   // 3 x (X tile + W tile) = 72 KB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, px = lane & 31;

    // workgroup id -> (row tile, column group): id%8 is the XCD the hardware dispatches to
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int cg = j % a.ncg;
    const int tile = (j / a.ncg) * 8 + xcd;
    if (tile >= a.ntiles) return;
    const int row0 = tile * TM;
    const int nkb = a.K >> 4;
    // this lane's global sources: X pieces (plane, k-half) of row row0+tid, W pieces tid and tid+256.
    // Rows past the end re-read the last row; their results are never stored.
    const int myrow = min(row0 + tid, a.M - 1);
    const __bf16* xa = a.x + (size_t)myrow * 8;          // piece q of this row: + q*M*8
    const f32x4* wb = reinterpret_cast<const f32x4*>(a.w) + (size_t)cg * nkb * B_Q + tid;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // asynchronous global -> LDS copies (global_load_lds_dwordx4): the destination is wave-uniform
    // base + lane*16, which is exactly the [piece][row] order of the tile
    // copy c (0..5) of K step kb into buffer buf: 0..3 = X pieces (plane, k-half), 4..5 = the two halves of the W tile
    auto issue_one = [&](int kb, int buf, int c) {
        f32x4* dst = lds + buf * (A_Q + B_Q) + wave * 64;
        if (c < 4) {
            const int p = c >> 1, h = c & 1;
            __builtin_amdgcn_global_load_lds((const void*)(xa + p * a.x_plane + (size_t)(kb * 2 + h) * a.M * 8),
                                             (lds_ptr)(dst + (p * 2 + h) * TM), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((const void*)(wb + (size_t)kb * B_Q + (c - 4) * 256), (lds_ptr)(dst + A_Q + (c - 4) * 256), 16, 0, 0);
        }
    };
    auto issue = [&](int kb, int buf) {
#pragma unroll
        for (int c = 0; c < 6; ++c) issue_one(kb, buf, c);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][s][r] = 0.f;

    // Three LDS buffers, copies two K steps ahead (6 copies per wave per step), and the FRAGMENTS one step ahead in registers
    // (round 3): the twelve ds_read_b128 of step kb+1 are issued behind the barrier that publishes it and land while the 24
    // MFMAs of step kb run.  Before, every step began with the reads and an `lgkmcnt(0)`: ~350 cycles of LDS latency in front
    // of 768 cycles of MFMAs, hidden only as far as the CU's second workgroup happened to be out of phase
    // (0.106 ms for the bare loop against 0.056 ms of MFMA time).
    const bool no_mma = KRK_DBGBIT(a, 1), no_copy = KRK_DBGBIT(a, 2), no_lds = KRK_DBGBIT(a, 8);
    const bool spread = a.nbuf != 2;          // probe (KRK_GEMM_SPREAD=0 -> nbuf 2): all six copies in front of the MFMAs, as before
    if (!no_copy) { issue(0, 0);
    if (nkb > 1) issue(1, 1); }

    const int arow = wave * 64 + px;   // first segment's row inside the tile
    struct Frag { bf16x8 xh[2], xl[2], wh[4], wl[4]; };
    auto read = [&](int buf, Frag& f) {
        const f32x4* L = lds + buf * (A_Q + B_Q);
        if (no_lds) return;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f.xh[s] = __builtin_bit_cast(bf16x8, L[(0 + half) * TM + arow + 32 * s]);
            f.xl[s] = __builtin_bit_cast(bf16x8, L[(2 + half) * TM + arow + 32 * s]);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f.wh[cb] = __builtin_bit_cast(bf16x8, L[A_Q + (0 + half) * TN + cb * 32 + px]);
            f.wl[cb] = __builtin_bit_cast(bf16x8, L[A_Q + (2 + half) * TN + cb * 32 + px]);
        }
    };
    // `refill_kb` >= 0: the six copies of that K step are issued BETWEEN the MFMAs (round 4): a copy instruction stands ~200-300
    // cycles in the CU's address path when all eight waves of the CU issue theirs (phase accounting of gemm_x3w.hip,
    // profiles/r04_phase_stats.txt); in front of the MFMAs that is dead time for the wave, behind a group of six MFMAs most of it
    // is covered by the matrix pipe working them off
    auto mma = [&](const Frag& f, int refill_kb, int refill_buf) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            if (!no_mma) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh[cb], f.xh[s], acc[cb][s], 0, 0, 0);
                    KRK_CROSS(acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wh[cb], f.xl[s], acc[cb][s], 0, 0, 0);
                              acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.wl[cb], f.xh[s], acc[cb][s], 0, 0, 0);)
                }
            }
            if (refill_kb >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (cb < 2) { issue_one(refill_kb, refill_buf, 2 * cb); issue_one(refill_kb, refill_buf, 2 * cb + 1); }
                else issue_one(refill_kb, refill_buf, 2 + cb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // one K step: publish step kb+1 (its copies landed, everyone is done reading step kb-1 ... kb), start its fragment reads,
    // refill the buffer step kb used, then the MFMAs of step kb on the fragments read one step ago
    auto step = [&](int kb, const Frag& cur, Frag& nxt) {
        if (kb + 1 < nkb) {
            if (kb + 2 < nkb) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // lgkmcnt(0) through the BUILTIN (vmcnt 63, expcnt 7, lgkmcnt 0): the compiler's own wait-count pass must know that
            // the fragments of step kb have arrived, or it waits for them at the first MFMA -- behind the reads of step kb+1
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            read((kb + 1) % 3, nxt);
            if (kb + 3 < nkb && !no_copy && !spread) issue(kb + 3, kb % 3);
        }
        // ONE call site for the MFMAs (two, with the accumulators flowing through both, made the compiler spill them)
        mma(cur, (spread && !no_copy && kb + 3 < nkb) ? kb + 3 : -1, kb % 3);
    };
    Frag fa = {}, fb = {};
    if (nkb > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read(0, fa);
    if (nkb > 2 && !no_copy) issue(2, 2);
    int kb = 0;
    for (; kb + 1 < nkb; kb += 2) {
        step(kb, fa, fb);
        step(kb + 1, fb, fa);
    }
    if (kb < nkb) step(kb, fa, fb);

    // epilogue: D[column][row] (weights are the MFMA's A operand): lane = row px of its segment, registers
    // 4j..4j+3 = columns 8j + 4*half + 0..3 of the block.  The tile is transposed through LDS (the pipeline
    // buffers are free now) so that every global store covers two full 512-byte row runs.
    __builtin_amdgcn_s_barrier();                      // slower waves may still be reading the last K step
    constexpr int RSTR = TN + 4;                       // floats per LDS row: +16 B keeps the column writes conflict-free
    float* T = reinterpret_cast<float*>(lds) + wave * (32 * RSTR);
    const int col0 = cg * TN;
    const bool vec = (a.Cout & 3) == 0;
    const bool nostore = KRK_DBGBIT(a, 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col0 + cb * 32 + 8 * j + 4 * half);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[cb][s][4 * j + i] + bv[i];
                *reinterpret_cast<f32x4*>(T + px * RSTR + cb * 32 + 8 * j + 4 * half) = v;
            }
        const int rbase = row0 + wave * 64 + 32 * s;
        int ln = 0, tt = 0;                                // (line, step) of row rbase + half (tile-time-major output only)
        int tl = 0, rem = 0;                               // (16-line tile, row inside it) of row rbase + half (tile-time-major input)
        const int T16 = 16 * (a.tileT < 0 ? -a.tileT : 1);
        if (a.tileT > 0) {
            ln = (rbase + half) / a.tileT;
            tt = (rbase + half) - ln * a.tileT;
        } else if (a.tileT < 0) {
            tl = (rbase + half) / T16;
            rem = (rbase + half) - tl * T16;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 2 * i + half;
            const f32x4 v = *reinterpret_cast<const f32x4*>(T + r * RSTR + 4 * px);
            const int row = rbase + r, col = col0 + 4 * px;
            size_t orow = (size_t)row;
            bool keep = true;
            if (a.tileT > 0) {
                orow = ((size_t)(ln >> 4) * a.tileT + tt) * 16 + (ln & 15);
                tt += 2;
                while (tt >= a.tileT) { tt -= a.tileT; ++ln; }
            } else if (a.tileT < 0) {
                const int n = tl * 16 + (rem & 15);
                keep = n < a.nlines;
                orow = (size_t)n * (size_t)(-a.tileT) + (rem >> 4);
                rem += 2;
                if (rem >= T16) { rem -= T16; ++tl; }
            }
            if (row < a.M && keep && !nostore) {
                float* yp = a.y + orow * a.Cout + col;
                if (vec) {
                    if (col < a.Cout) {
                        *reinterpret_cast<f32x4*>(yp) = v;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < a.Cout) yp[e] = v[e];
                }
            }
        }
    }
}

}  // namespace

int KRK_FN(krk_launch_gemm_x3)(const GemmX3Args& a, hipStream_t s) {
    if (a.K % 16 || a.M <= 0) return a.M == 0 ? 0 : -1;
    const int slots = (a.ntiles + 7) / 8 * 8;
    const size_t lds = (size_t)3 * (A_Q + B_Q) * 16;
    // the attribute belongs to the function object of the CURRENT device: once per device, not once per process
    static bool attr_set[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(gemm_x3_kernel, dim3((unsigned)(slots * a.ncg)), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
