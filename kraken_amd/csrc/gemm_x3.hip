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
#include <stdlib.h>

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

// ------------------------------------------------------------------------------------------------------------------------
// Round 6: the PERSISTENT form (gemm_x3p_kernel).  What the one-tile-per-workgroup kernel above cannot do: overlap a tile's
// 128 KB of fp32 row stores with matrix work.  Its two co-resident workgroups start together, and they STAY in phase -- a
// workgroup that lags finds the matrix pipe to itself while its neighbour stores and catches up, so every stagger decays by
// half per tile (which is why round 2's staggered start measured nothing) -- and the kernel paid 0.160 ms where its K loops
// take 0.123.  Here ONE workgroup of eight waves owns a CU for the whole launch:
//   tile      = the same 256 rows x 128 columns, the same LDS stages, the same MFMA order per accumulator (bit-identical
//               results), but a wave holds 64 x 64 of it (64 accumulator registers), so that the FINISHED tile fits next to
//               the running one: at a tile's end the accumulators move to `prev`, and `prev` is drained -- + bias, transposed
//               through a per-wave LDS scratch into whole 256-byte row runs, stored -- in seventeen slices that ride on K steps
//               1..17 of the NEXT tile.  The stores are raw-buffer stores that are ALWAYS issued (out-of-range offset = dropped):
//               every hand-counted vmcnt below knows exactly how many are in flight.
//   tiles     = claimed at run time from eight queues (one per XCD: the column groups of a row tile run back to back on one
//               L2, as before; an XCD that runs dry steals from the next).  The claim for tile i+1 is an atomic issued at the end
//               of tile i's K step 0 and read at the end of step 2; its tile's first copies are needed at step nkb-3.
//   pipeline  = the three LDS stages run straight through the tile boundaries (a tile's last K steps already copy the next
//               tile's first ones): no drain and refill per tile.
//   grid      = one workgroup per CU (or per tile if there are fewer); workgroups that start late -- their CU was busy with
//               another batch's recurrence -- find the queues empty and leave.
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int P_STAGE = (A_Q + B_Q) * 16;            // bytes per pipeline stage: 24 KB (16 KB of X, 8 KB of W)
constexpr int P_SCR_ROW = (64 + 4) * 4;              // bytes per scratch row: 64 columns + 16 B (conflict-free column writes)
constexpr int P_SCR_WAVE = 32 * P_SCR_ROW;           // 8704 bytes per wave
constexpr int P_OFF_SCR = 3 * P_STAGE;
constexpr int P_OFF_BIAS = P_OFF_SCR + 8 * P_SCR_WAVE;
constexpr int P_BIAS_MAX = 2048;                     // floats: Cout rounded up to 128 must fit
constexpr int P_OFF_CLAIM = P_OFF_BIAS + P_BIAS_MAX * 4;
constexpr int P_LDS = P_OFF_CLAIM + 64;              // 151 616 bytes: one workgroup per CU
constexpr int P_UNROLL = 20;                         // K steps of a tile written out (the drain rides on steps 1..17)
constexpr unsigned P_OOB = 0x80000000u;              // voffset beyond the descriptor (outputs < 2 GiB): the store is dropped

// stores issued by the drain slice that rides on K step j (see drain()): two in each read slice but the first, two in the flush
constexpr int p_stores(int j) { return ((j >= 6 && j <= 9) || (j >= 14 && j <= 17)) ? 2 : 0; }
// vmcnt at the top of K step kb: its stage's three copies were issued in step kb-2; behind them in the queue are that step's
// stores, the three copies of step kb-1 and its stores
constexpr int p_wait(int kb) { return 3 + p_stores(kb - 2) + p_stores(kb - 1); }

template <int N>
__device__ __forceinline__ void p_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

template <int V> struct PTag { static constexpr int value = V; };
struct PFrag { bf16x8 xh[2], xl[2], wh[2], wl[2]; };

// MODE: the row order of the output (GemmX3Args::tileT): 0 plain, 1 line-major in -> tile-time-major out, -1 the reverse
template <int MODE>
__global__ void __launch_bounds__(512, 1) gemm_x3p_kernel(const GemmX3Args a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];   // dynamic LDS starts at address 0 (no static LDS in this kernel)
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, px = lane & 31;
    const int rq = wave & 3, ch = wave >> 2;          // the wave's 64 rows and 64 columns of the tile
    const int nkb = a.K >> 4;
    const int xcd = blockIdx.x & 7;
    float* const biasL = reinterpret_cast<float*>(ldsb + P_OFF_BIAS);
    int* const claimL = reinterpret_cast<int*>(ldsb + P_OFF_CLAIM);   // [0] the claimed tile, [1] thread 0's queue scan position (plain LDS: every reader sits behind a wait with a memory clobber)
    for (int i = tid; i < a.ncg * TN; i += 512) biasL[i] = i < a.Cout ? a.bias[i] : 0.f;

    // ---- tile queues (thread 0).  Queue x holds the row tiles x, x+8, ... times all column groups, column group fastest.
    // claimL[1] ("scan"): queues (xcd + scan) & 7 and later may still hold tiles
    auto queue_len = [&](int x) { return ((a.ntiles - x + 7) >> 3) * a.ncg; };
    auto tile_id = [&](int x, int q) { const int r = q / a.ncg; return (r * 8 + x) * a.ncg + (q - r * a.ncg); };   // row tile * ncg + column group
    auto claim_now = [&](int& scan) -> int {
        while (scan < 8) {
            const int x = (xcd + scan) & 7;
            const int q = (int)atomicAdd(a.ctr + x, 1u);
            if (q < queue_len(x)) return tile_id(x, q);
            ++scan;
        }
        return -1;
    };
    if (tid == 0) { int sc = 0; claimL[0] = claim_now(sc); claimL[1] = sc; }
    __syncthreads();
    int cur_id = __builtin_amdgcn_readfirstlane(claimL[0]);
    __builtin_amdgcn_s_barrier();                     // claimL[0] is rewritten during the first tile

    // ---- copy sources: a 64-bit wave-uniform base (SGPRs) + a 32-bit lane offset.  X piece (plane, k-half = ch) of row chunk rq:
    // only the lane offset (the row) depends on the tile; W chunk `wave`: only the base (the column group) does.
    const size_t xstep = (size_t)a.M * 32;            // bytes between K steps of one k-half (two 8-element K blocks)
    const char* const xb0 = reinterpret_cast<const char*>(a.x) + (size_t)ch * a.M * 16;
    const char* const xb1 = xb0 + a.x_plane * 2;
    const char* const wb0 = reinterpret_cast<const char*>(a.w) + wave * 1024;
    const unsigned wo = (unsigned)lane * 16;
    auto x_off = [&](int id) -> unsigned { return (unsigned)min((id / a.ncg) * TM + rq * 64 + lane, a.M - 1) * 16u; };
    const unsigned dX0 = (unsigned)(((0 + ch) * TM + rq * 64) * 16), dX1 = (unsigned)(((2 + ch) * TM + rq * 64) * 16);
    const unsigned dW = (unsigned)((A_Q + wave * 64) * 16);
    auto p_copy = [&](const char* sbase, unsigned voff, unsigned lds_off) {      // 64 lanes x 16 B -> LDS [lds_off, +1 KB)
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
    };
    // the three copies of K step `kb` of the tile (row offset xo, column group cg) into stage b
    auto copy3 = [&](unsigned xo, int cg, int kb, int b) {
        const unsigned st = (unsigned)(b * P_STAGE);
        p_copy(xb0 + (size_t)kb * xstep, xo, st + dX0);
        p_copy(xb1 + (size_t)kb * xstep, xo, st + dX1);
        p_copy(wb0 + ((size_t)cg * nkb + kb) * (B_Q * 16), wo, st + dW);
    };

    f32x16 acc[2][2], prev[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[cb][s][r] = 0.f; prev[cb][s][r] = 0.f; }

    if (cur_id >= 0) {
        unsigned cu_xo = x_off(cur_id), nx_xo = cu_xo;
        int cu_cg = cur_id % a.ncg, nx_cg = cu_cg;
        int nxt_id = -1;
        int prev_rt = 0, prev_cg = 0;
        bool prev_valid = false;
        int qv = 0;                                    // thread 0: the claim in flight

        // ---- drain state: output descriptor, scratch, row iterator of the segment being stored
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0x7FFFFFFF, 0x00020000);
        char* const scr = ldsb + P_OFF_SCR + wave * P_SCR_WAVE;
        const bool nostore = KRK_DBGBIT(a, 4);
        const int T = __builtin_amdgcn_readfirstlane(MODE < 0 ? -a.tileT : a.tileT);     // steps per line (MODE != 0)
        const int lim = MODE > 0 ? T : 16 * T;
        int ia = 0, ib = 0, irow = 0;                  // row iterator: (line, step) / (16-line tile, row in it) / plain; irow = the input row
        f32x4 pend[2];
        unsigned pend_off[2] = {P_OOB, P_OOB};
        pend[0] = pend[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto row_init = [&](int R) {
            irow = R;
            if constexpr (MODE != 0) { ia = R / lim; ib = R - ia * lim; }
        };
        auto row_adv4 = [&]() {
            irow += 4;
            if constexpr (MODE != 0) {                 // four corrections cover every T >= 1
                ib += 4;
                bool w;
                w = ib >= lim; ib -= w ? lim : 0; ia += w ? 1 : 0;
                w = ib >= lim; ib -= w ? lim : 0; ia += w ? 1 : 0;
                w = ib >= lim; ib -= w ? lim : 0; ia += w ? 1 : 0;
                w = ib >= lim; ib -= w ? lim : 0; ia += w ? 1 : 0;
            }
        };
        auto row_off = [&](int col) -> unsigned {
            bool keep = prev_valid && irow < a.M && col < a.Cout && !nostore;
            unsigned orow = (unsigned)irow;            // the launcher admits outputs below 2 GiB only: 32-bit row arithmetic
            if constexpr (MODE > 0) orow = ((unsigned)(ia >> 4) * (unsigned)T + (unsigned)ib) * 16u + (unsigned)(ia & 15);
            else if constexpr (MODE < 0) {
                const int n = ia * 16 + (ib & 15);
                keep = keep && n < a.nlines;
                orow = (unsigned)n * (unsigned)T + (unsigned)(ib >> 4);
            }
            const unsigned off = (orow * (unsigned)a.Cout + (unsigned)col) * 4u;
            return keep ? off : P_OOB;
        };
        // slice D (0..16) of the drain of `prev`: per 32-row segment s four WRITE slices (two (column block, j) quads each: + bias,
        // column order -> scratch rows), then four READ slices (two row quads each: 4 rows x 256 B per instruction); what a read
        // slice fetched is stored by the NEXT slice, so no slice waits for its own LDS reads
        auto drain = [&](auto dtag) {
            constexpr int D = decltype(dtag)::value;
            if constexpr (D >= 0 && D <= 16) {
                constexpr int s = (D >> 3) & 1, ph = D & 7;
                if constexpr (D == 16 || ph >= 5 || D == 8) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pend[e]), yrs, (int)pend_off[e], 0, 0);
                }
                if constexpr (D < 16 && ph < 4) {
                    if constexpr (ph == 0) row_init(prev_rt * TM + rq * 64 + s * 32 + (lane >> 4));
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        constexpr int u = 2 * ph;
                        const int cb = (u + e) >> 2, j = (u + e) & 3;
                        const int c = cb * 32 + 8 * j + 4 * half;
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(biasL + prev_cg * TN + ch * 64 + c);
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = prev[cb][s][4 * j + i] + bv[i];
                        *reinterpret_cast<f32x4*>(scr + px * P_SCR_ROW + c * 4) = v;
                    }
                } else if constexpr (D < 16) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int i2 = (ph - 4) * 2 + e;
                        pend[e] = *reinterpret_cast<const f32x4*>(scr + (4 * i2 + (lane >> 4)) * P_SCR_ROW + (lane & 15) * 16);
                        pend_off[e] = row_off(prev_cg * TN + ch * 64 + 4 * (lane & 15));
                        row_adv4();
                    }
                }
            }
        };

        // ---- fragments and MFMAs of one K step
        const int arow = rq * 64 + px;
        auto read = [&](int b, PFrag& f) {
            const f32x4* L = reinterpret_cast<const f32x4*>(ldsb + b * P_STAGE);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f.xh[s] = __builtin_bit_cast(bf16x8, L[(0 + half) * TM + arow + 32 * s]);
                f.xl[s] = __builtin_bit_cast(bf16x8, L[(2 + half) * TM + arow + 32 * s]);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f.wh[cb] = __builtin_bit_cast(bf16x8, L[A_Q + (0 + half) * TN + ch * 64 + cb * 32 + px]);
                f.wl[cb] = __builtin_bit_cast(bf16x8, L[A_Q + (2 + half) * TN + ch * 64 + cb * 32 + px]);
            }
        };
        int b = 0;                                     // stage of the step about to run
        // one K step.  NW = vmcnt that proves the NEXT step's stage has landed; D = drain slice riding on this step
        auto step = [&](auto wtag, auto dtag, int kb, const PFrag& cur, PFrag& nxt) {
            p_vmwait<decltype(wtag)::value>();
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0) through the builtin: the compiler's own wait counting must see it
            __builtin_amdgcn_s_barrier();
            const int bn = b == 2 ? 0 : b + 1;
            read(bn, nxt);
            // the copies of step kb+3 go to the stage this step's fragments came from (every wave is past reading it)
            const int k3 = kb + 3;
            const bool own = k3 < nkb;
            const unsigned xo = own ? cu_xo : nx_xo;
            const int cgc = own ? cu_cg : nx_cg;
            const int kc = own ? k3 : k3 - nkb;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.wh[cb], cur.xh[s], acc[cb][s], 0, 0, 0);
                    KRK_CROSS(acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.wh[cb], cur.xl[s], acc[cb][s], 0, 0, 0);
                              acc[cb][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.wl[cb], cur.xh[s], acc[cb][s], 0, 0, 0);)
                }
                if (cb == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    copy3(xo, cgc, kc, b);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            drain(dtag);
            b = bn;
        };

        // ---- prologue: the first tile's K steps 0..2 into the three stages
#pragma unroll
        for (int k = 0; k < 3; ++k) copy3(cu_xo, cu_cg, k, k);
        PFrag fa = {}, fb = {};
        p_vmwait<6>();
        __builtin_amdgcn_s_barrier();
        read(0, fa);

        while (true) {
            // K steps 0..19 written out: the previous tile's drain rides on steps 1..17, the claim of the next tile on 0..4
            step(PTag<p_wait(0)>{}, PTag<-1>{}, 0, fa, fb);
            if (tid == 0) {
                // The claim of the next tile, in flight during steps 1 and 2.  By hand: hipcc's atomic optimizer turns atomicAdd into
                // "one lane adds, s_waitcnt vmcnt(0), v_readfirstlane" on the spot -- a full round trip to L2 in front of every
                // wave of the workgroup.  The result register must not be touched before the wait at the end of step 2.
                const int sc = claimL[1];
                const unsigned qoff = sc < 8 ? (unsigned)((xcd + sc) & 7) * 4u : 36u, one = 1u;     // [9]: a counter nobody reads
                asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(qv) : "v"(qoff), "v"(one), "s"(a.ctr) : "memory");
            }
            step(PTag<p_wait(1)>{}, PTag<0>{}, 1, fb, fa);
            step(PTag<p_wait(2)>{}, PTag<1>{}, 2, fa, fb);
            if (tid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(qv) : : "memory");     // wave 0 only: the claim (and this wave's copies so far)
                int sc = claimL[1];
                const int qx = (xcd + sc) & 7;
                int id;
                if (sc < 8 && qv < queue_len(qx)) id = tile_id(qx, qv);
                else { if (sc < 8) ++sc; id = claim_now(sc); claimL[1] = sc; }
                claimL[0] = id;
            }
            step(PTag<p_wait(3)>{}, PTag<2>{}, 3, fb, fa);
            step(PTag<p_wait(4)>{}, PTag<3>{}, 4, fa, fb);
            nxt_id = __builtin_amdgcn_readfirstlane(claimL[0]);      // written before the barrier of step 4
            if (nxt_id >= 0) { nx_xo = x_off(nxt_id); nx_cg = nxt_id % a.ncg; }
            else { nx_xo = cu_xo; nx_cg = cu_cg; }                  // no next tile: the last steps copy (harmlessly) from this one
            step(PTag<p_wait(5)>{}, PTag<4>{}, 5, fb, fa);
            step(PTag<p_wait(6)>{}, PTag<5>{}, 6, fa, fb);
            step(PTag<p_wait(7)>{}, PTag<6>{}, 7, fb, fa);
            step(PTag<p_wait(8)>{}, PTag<7>{}, 8, fa, fb);
            step(PTag<p_wait(9)>{}, PTag<8>{}, 9, fb, fa);
            step(PTag<p_wait(10)>{}, PTag<9>{}, 10, fa, fb);
            step(PTag<p_wait(11)>{}, PTag<10>{}, 11, fb, fa);
            step(PTag<p_wait(12)>{}, PTag<11>{}, 12, fa, fb);
            step(PTag<p_wait(13)>{}, PTag<12>{}, 13, fb, fa);
            step(PTag<p_wait(14)>{}, PTag<13>{}, 14, fa, fb);
            step(PTag<p_wait(15)>{}, PTag<14>{}, 15, fb, fa);
            step(PTag<p_wait(16)>{}, PTag<15>{}, 16, fa, fb);
            step(PTag<p_wait(17)>{}, PTag<16>{}, 17, fb, fa);
            step(PTag<p_wait(18)>{}, PTag<-1>{}, 18, fa, fb);
            step(PTag<p_wait(19)>{}, PTag<-1>{}, 19, fb, fa);
            static_assert(P_UNROLL == 20 && p_wait(20) == 3 && p_wait(21) == 3, "the run-time K loop below waits vmcnt(3)");
            int kb = P_UNROLL;
            for (; kb + 1 < nkb; kb += 2) {
                step(PTag<3>{}, PTag<-1>{}, kb, fa, fb);
                step(PTag<3>{}, PTag<-1>{}, kb + 1, fb, fa);
            }
            if (kb < nkb) {
                step(PTag<3>{}, PTag<-1>{}, kb, fa, fb);
                fa = fb;
            }
            // the tile is complete: park it, start the next one
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    prev[cb][s] = acc[cb][s];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[cb][s][r] = 0.f;
                }
            prev_rt = cur_id / a.ncg;
            prev_cg = cur_id - prev_rt * a.ncg;
            prev_valid = true;
            if (nxt_id < 0) break;
            cur_id = nxt_id;
            cu_xo = nx_xo;
            cu_cg = nx_cg;
        }
        // the last tile's drain has nothing to ride on (the dummy copies of the last three steps may still be landing: the
        // scratch region is not theirs)
        drain(PTag<0>{}); drain(PTag<1>{}); drain(PTag<2>{}); drain(PTag<3>{}); drain(PTag<4>{}); drain(PTag<5>{});
        drain(PTag<6>{}); drain(PTag<7>{}); drain(PTag<8>{}); drain(PTag<9>{}); drain(PTag<10>{}); drain(PTag<11>{});
        drain(PTag<12>{}); drain(PTag<13>{}); drain(PTag<14>{}); drain(PTag<15>{}); drain(PTag<16>{});
        p_vmwait<0>();                                 // LDS-DMA copies must not land in a successor workgroup's LDS
    }
    // the last workgroup out zeroes the queues for the next launch on this stream
    if (tid == 0) {
        const unsigned gone = atomicAdd(a.ctr + 8, 1u);
        if (gone == gridDim.x - 1) {
#pragma unroll
            for (int x = 0; x < 10; ++x) atomicExch(a.ctr + x, 0u);
        }
    }
}

}  // namespace

int KRK_FN(krk_launch_gemm_x3)(const GemmX3Args& a, hipStream_t s) {
    if (a.K % 16 || a.M <= 0) return a.M == 0 ? 0 : -1;
    static bool attr_set[64] = {false};
    static int ncu[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool known = dev >= 0 && dev < 64;
    // the attribute belongs to the function object of the CURRENT device: once per device, not once per process
    if (!known || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)3 * (A_Q + B_Q) * 16));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3p_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3p_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3p_kernel<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, known ? dev : 0);
        if (known) { attr_set[dev] = true; ncu[dev] = n; }
    }
    // persistent form: enough K steps to carry a tile's drain, 16-byte row pieces, bias and output inside its LDS / descriptor
    const int nkb = a.K >> 4;
    const size_t out_rows = (size_t)a.M + 16 * (size_t)(a.tileT < 0 ? -a.tileT : a.tileT);
    int cus = known && ncu[dev] > 0 ? ncu[dev] : 256;
    static const int grid_probe = [] { const char* e = getenv("KRK_GEMM_PGRID"); return e ? atoi(e) : 0; }();   // A/B probe: workgroups of the persistent grid
    if (grid_probe > 0) cus = grid_probe;
    const long tiles = (long)a.ntiles * a.ncg;
    // at least four tiles per workgroup: below that the dynamic queue's tail (one tile = 1/4 of the launch) costs more than the
    // overlapped stores save (LinSoftmax with 256 classes: 300 tiles -- 0.20 ms here against 0.065 ms one tile per workgroup)
    if (a.ctr && nkb >= P_UNROLL && (a.Cout & 3) == 0 && a.ncg * TN <= P_BIAS_MAX && out_rows * a.Cout * 4 < 0x7FFFFFFFull &&
        tiles >= 4L * cus && !(KRK_DBGBIT(a, 1) || KRK_DBGBIT(a, 2) || KRK_DBGBIT(a, 8))) {
        if (a.tileT > 0) hipLaunchKernelGGL(gemm_x3p_kernel<1>, dim3((unsigned)cus), dim3(512), P_LDS, s, a);
        else if (a.tileT < 0) hipLaunchKernelGGL(gemm_x3p_kernel<-1>, dim3((unsigned)cus), dim3(512), P_LDS, s, a);
        else hipLaunchKernelGGL(gemm_x3p_kernel<0>, dim3((unsigned)cus), dim3(512), P_LDS, s, a);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const int slots = (a.ntiles + 7) / 8 * 8;
    const size_t lds = (size_t)3 * (A_Q + B_Q) * 16;
    hipLaunchKernelGGL(gemm_x3_kernel, dim3((unsigned)(slots * a.ncg)), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
