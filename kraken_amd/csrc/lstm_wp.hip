// Pipelined weight-stationary recurrent kernel of the (bi)directional LSTM (round 3; successor of lstm_ws.hip for the
// shapes it covers).  Split-bf16 operands on the bf16 matrix cores, W_hh resident in REGISTERS of a cluster of CS
// workgroups (one per CU) that exchange h_t through L2 with data-tagged granules.
// Reference semantics: nn.LSTM inside TransposedSummarizingRNN.forward (kraken/lib/vgsl/layers.py:513-547): packed by
// length, gates i,f,g,o, h/c start at zero, outputs past a line's length stay zero.
//
// What lstm_ws.hip measured (profiles/r02_lstm_ws_pmc.txt): a slot was a latency CHAIN -- barrier, 14 KB of fragment reads per
// wave, 42 MFMAs, then the gate math (exposed for the last block), publish, gather check, barrier -- 1.57 us of which the matrix
// pipe worked 0.56 us, and every wave took part in every phase at the same time.  Here the chain is cut in three places:
//   * GATES ARE DEFERRED BY ONE STAGE.  A stage = one time step of one 16-line group.  The pre-activations z(g, s) of a stage
//     stay in four registers; the gate math, the cell update, the h rows written to LDS and the published granules of stage k
//     are issued INSIDE the MFMA stream of stage k+1 (another group: independent data), one MFMA : ~2 VALU, so the VALU work
//     runs in the shadow of the matrix pipe instead of after it.  NG = 4 groups per cluster: h(g, s) leaves in the first half
//     of stage k+1 and is needed by stage k+4;
//   * THE EXCHANGE HAS ITS OWN WAVES.  vmcnt retires in order: a wave that publishes (a store, ~1 us to be acknowledged)
//     and streams xproj from HBM cannot also wait for gather loads without waiting for those.  Waves 8..11 do nothing but
//     gather: poll the peers' granules of the group the NEXT stage consumes until every tag matches, drop the payloads into
//     the LDS rows, arrive at the stage barrier.  No optimistic path, no flags, no slow path in the compute waves; a late
//     granule simply holds the barrier.  The compute waves never wait for anything younger than two stages: xproj lands in
//     LDS (global_load_lds) THREE stages ahead and is waited for with a fixed count (every stage issues exactly
//     xproj load, output store, publish store);
//   * CS = 8 SLICES, ONE BLOCK PER WAVE.  A wave owns one block of 16 gate columns (all four gates of four units): 56 weight
//     registers, 21 MFMAs per stage (N = 256: 8 clusters x 8 CUs; H <= 128: CS = 4).  The plan was a stage of ~0.45 us against
//     the 1.5 of lstm_ws.hip's two-block slot; measured: 0.78 us -- half the work in half the time, the same chip time.
// Cluster membership is claimed at run time, per XCD (see the claim below: a partially resident grid cannot deadlock), spins are
// bounded and raise the plan's status word, h lives in LDS as [plane][K octet][line][K block] (conflict-free fragment reads), ONE
// buffer per group; granules are 16 bytes of payload with a 4-bit step tag in the lo parts' lowest mantissa bits, written with one
// plain store into a buffer the host zeroes before every launch, two parity slots per group; the output pass (tile-time-major or
// line-major rows) reads the gathered rows.  What the round-3 measurements say about this design -- it equals lstm_ws.hip, it does
// not beat it -- is in DESIGN.md 3.3 and profiles/r03_lstm_wp_gather_variants.txt.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// timeline stamps (ablation build only): stages [64, 96) of cluster 0 / slice 0, lane 0 of compute wave 0 (slots 0..3) and of
// gather wave 8 (slots 4..7): s_memtime into LDS -- NOT into memory: a store would sit in the wave's vmcnt queue and the next
// hand-counted wait would wait for it (the first timelines of round 3 showed a 1200-cycle "poll" that was the stamp's own
// store) -- copied to tl[(stage - 64) * 8 + slot] when the wave ends
// Slots per stage (32): 0..3 compute wave 0 (barrier exit | operands issued | MFMAs done | next barrier entry), 4..7 gather wave 8
// (barrier exit | first poll back | tags ok | next barrier entry), 8 + w: barrier entry of wave w, 20 + w: barrier exit of wave w.
#ifdef KRK_ABLATE
#define WP_STAMP(cond, kk, slot) do { if ((cond) && a.tl && cluster == 0 && slice == 0 && lane == 0 && (kk) >= 64u && (kk) < 96u) \
    reinterpret_cast<unsigned long long*>(smem8 + tl_off)[((kk) - 64u) * 32u + (slot)] = __builtin_readcyclecounter(); } while (0)
#define WP_STAMPS_OUT() do { if (a.tl && cluster == 0 && slice == 0) { __syncthreads(); \
    for (int q_ = tid; q_ < 1024; q_ += 768) a.tl[q_] = reinterpret_cast<unsigned long long*>(smem8 + tl_off)[q_]; } } while (0)
#else
#define WP_STAMP(cond, kk, slot) do {} while (0)
#define WP_STAMPS_OUT() do {} while (0)
#endif

namespace {

constexpr unsigned kOOBwp = 0x80000000u;   // voffset beyond every descriptor used here (all < 2 GiB): load = 0, store dropped

__device__ __forceinline__ bf16x8 wp_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 wp_srd(const void* p, unsigned bytes) {       // raw buffer descriptor in SGPRs
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    i32x4 r;
    r[0] = (int)__builtin_amdgcn_readfirstlane((unsigned)u);
    r[1] = (int)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xFFFFu);
    r[2] = (int)__builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}
// every vector-memory instruction of the time loop is inline assembly: the compiler neither sees nor counts them, so the only
// vmcnt waits are the hand-counted ones below
__device__ __forceinline__ void wp_load_b128_sc1(u32x4& d, unsigned vo, const i32x4& srd, unsigned so) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "=&v"(d) : "v"(vo), "s"(srd), "s"(so) : "memory");
}
// publish: a PLAIN store.  A cluster lives on ONE XCD (see the cluster claim), whose L2 is the point of coherence of its CUs:
// the store writes through the CU's L1 into that L2 and stays there; the peers' polls (sc1 loads: bypass L1, served by L2) hit
// it a few hundred cycles later.  (sc1 stores write through to the fabric and drop the line from L2: ~2 us per hop under load.)
// (16 bytes: see the hazard note at wp_store_b128)
__device__ __forceinline__ void wp_store_b128_so(const u32x4& d, unsigned vo, const i32x4& srd, unsigned so) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(d), "v"(vo), "s"(srd), "s"(so) : "memory");
}
// The s_nop is a HAZARD fix, not padding: a vector-memory store of more than 64 bits reads its data registers for a few cycles
// after issue, and the next VALU instruction must not write them (the hazard recogniser inserts the wait states for the
// compiler's own stores; it cannot see inside inline assembly).  The registers are dead for the compiler after this statement,
// so it reuses them at once -- round 3 bug: the first dword of a 16-byte output piece held the ADDRESS of the next stage's
// LDS read for the lanes whose data had not been fetched yet (lines 12..15 of a group).
__device__ __forceinline__ void wp_store_b128(const u32x4& d, unsigned vo, const i32x4& srd) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(d), "v"(vo), "s"(srd) : "memory");
}
__device__ __forceinline__ void wp_load_lds_b128(const float* gptr, unsigned lds_off) {   // 64 lanes x 16 B -> LDS [lds_off, +1 KB)
    unsigned keep;                                                                        // M0 (LDS base of the copy) is saved and restored
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_off) : "memory");
}
template <int N>
__device__ __forceinline__ void wp_vmwait() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// LDS-only workgroup barrier: vector memory stays in flight across it.  The explicit lgkmcnt(0) is NOT redundant: at a loop
// header hipcc (ROCm 7.2) drops the wait its own release fence needs when the LDS stores come from the loop's back edge --
// the gather waves' last ds_writes of an iteration were still in flight when the compute waves read the rows (round 3 bug:
// lines 12..15 of a group wrong, always the last lanes to land).
__device__ __forceinline__ void wp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// NKB K blocks of 32; NG groups of 16 lines per cluster (stages per time step); CS slices (workgroups) per cluster.
// 12 waves: 0..7 compute (one gate-column block each), 8..11 gather.
template <int NKB, int NG, int CS>
__global__ void __launch_bounds__(768) lstm_wp_kernel(const LstmWsArgs a) {
    constexpr int NPEER = CS - 1;
    constexpr int RING = 4;                                       // xproj landing buffers (three stages ahead)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    // h in LDS, per (group, parity): [plane hi|lo][K octet 4][line 16][K block: 16 bytes each, padded to an odd count]
    constexpr int RSO = 16 * (NKB | 1);     // bytes per (octet, line) row
    constexpr int OS = 16 * RSO;            // one octet plane
    constexpr int plane = 4 * OS;           // hi / lo plane
    constexpr int hbuf = 2 * plane;         // one (hi, lo) buffer of one group
    auto lds_of = [&](int ln, int unit) -> unsigned {        // byte offset of (line, unit) inside a buffer's hi plane
        return (unsigned)(((unit & 31) >> 3) * OS + ln * RSO + (unit >> 5) * 16 + (unit & 7) * 2);
    };
    unsigned char* hs = smem8;              // [group NG][hbuf]: ONE buffer per group (see the hazard note at the stage loop)
    int* lens_s = reinterpret_cast<int*>(smem8 + NG * hbuf);            // [16 * NG]
    unsigned* misc = reinterpret_cast<unsigned*>(lens_s + 16 * NG);     // [0] cluster (work item), [1] slice
    const unsigned dump_base = (unsigned)(NG * hbuf + 16 * NG * 4 + 16);   // masked LDS writes: 16 dump bytes per lane
    const unsigned xs_off = dump_base + 768u * 16u;                      // xproj landing ring [RING][wave 8][64 lanes x 16 B]
    [[maybe_unused]] const unsigned tl_off = xs_off + (unsigned)RING * 8u * 1024u;   // 8 KB of timeline stamps (ablation build)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- cluster claim: CS workgroups that run on the SAME XCD.  Arrivals are counted per XCD (ctrl[x]); the workgroup that
    // completes a local group of CS claims a work item c (ctrl[8]) and posts it in the group's mailbox; the others wait for the
    // mailbox.  The grid holds 8 (CS - 1) workgroups more than the C clusters need, so that C full groups form whatever the
    // distribution of blocks over XCDs (sum_x floor(n_x / CS) >= (G - 8 (CS - 1)) / CS = C); with the round-robin placement
    // observed (block b on XCD b % 8) the surplus blocks find all work claimed and leave at once.  Progress needs only CS
    // co-resident workgroups on one XCD; a group that can never fill ends when all C items are claimed.
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned C = (unsigned)a.nclusters;
        const unsigned t = atomicAdd(a.ctrl + xcc, 1u);
        const unsigned lc = t / CS, sl = t % CS;
        unsigned* mb = a.ctrl + 16 + xcc * (unsigned)a.mbox + lc;
        unsigned c = C;
        if (lc < (unsigned)a.mbox) {
            if (sl == CS - 1) {
                c = atomicAdd(a.ctrl + 8, 1u);
                __hip_atomic_store(mb, c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ctrl[9] counts the POSTED work items: a waiting workgroup may only conclude "my group is surplus" from a
                // counter that is bumped after the mailboxes are visible (ctrl[8] is bumped before: a member of the group that
                // took the last item would see "all claimed" ahead of its own mailbox and leave its cluster short)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (c < C) atomicAdd(a.ctrl + 9, 1u);
            } else {
                unsigned spins = 0;
                while (true) {
                    unsigned v = __hip_atomic_load(mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!v && __hip_atomic_load(a.ctrl + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= C)
                        v = __hip_atomic_load(mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // every item is posted: look once more
                    else if (!v) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 22)) {      // seconds: the group never filled and the work was never claimed
                            __hip_atomic_store(a.err, 0x40000000u | (xcc << 8) | sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                        continue;
                    }
                    if (v) c = v - 1u;                   // else: all work is posted and none of it is ours -- surplus
                    break;
                }
            }
        }
        misc[0] = c;
        misc[1] = sl;
    }
    for (int e = tid; e < NG * hbuf / 4; e += 768) reinterpret_cast<unsigned int*>(hs)[e] = 0u;   // NG*hbuf bytes
    __syncthreads();
    const int cluster = (int)__builtin_amdgcn_readfirstlane(misc[0]), slice = (int)__builtin_amdgcn_readfirstlane(misc[1]);
    if (cluster >= a.nclusters) return;                          // surplus workgroup
    const int dir = cluster % a.ndir;
    const int n0 = (cluster / a.ndir) * 16 * NG;
    const bool rev = (a.dirmode == 1) || (a.dirmode == 2 && dir == 1);
    if (tid < 16 * NG) {
        const int n = n0 + tid;
        int l = 0;
        if (n < a.N) l = a.lens ? min(max(a.lens[n], 0), a.T) : a.T;
        lens_s[tid] = l;
    }
    __syncthreads();
    int Lmax = 0;
    for (int i = 0; i < 16 * NG; ++i) Lmax = max(Lmax, lens_s[i]);
    Lmax = __builtin_amdgcn_readfirstlane(Lmax);
    if (Lmax <= 0) return;

    const int BPC = a.BPC;                                        // blocks per slice (<= 8)
    const unsigned slice_gran = (unsigned)BPC * 16u;              // granules one slice publishes per (group, step)
    const unsigned gp_bytes = (unsigned)CS * slice_gran * 16u;    // bytes per (group, parity)
    // granules of this cluster: [group][parity][slice][block][line 16] x 16 bytes.  A granule is ALL payload: the four units of a
    // block for one line, {hi0 | hi1 << 16, hi2 | hi3 << 16, lo0 | lo1 << 16, lo2 | lo3 << 16} -- exactly the 8 + 8 bytes of the
    // (line, block) position in the hi and lo planes of the LDS rows.  Its sequence tag lives in the lowest mantissa bit of the
    // four lo values (the lo parts are rounded to 7 significant bits: h = hi + lo is carried to 2^-17 instead of 2^-18): four
    // bits, (step mod 15) + 1, never 0.  That is enough because the host zeroes the granule buffer before every launch and a
    // (group, parity) slot is only ever overwritten by the step two after the one it held (15 is odd: the tags differ).  Round 3
    // first shipped {payload, 32-bit tag} pairs: twice the bytes through the gather waves, which bound the stage.
    const i32x4 grs = wp_srd(reinterpret_cast<const unsigned char*>(a.gran) + (size_t)cluster * (2 * NG) * gp_bytes, (unsigned)(2 * NG) * gp_bytes);
    auto tagbits = [](int ts, unsigned& w2, unsigned& w3) {        // tag of step ts - 1 (ts >= 1), spread over bits 0 / 16 of dwords 2, 3
        const unsigned t = (unsigned)((ts - 1) % 15) + 1u;
        w2 = (t & 1u) | ((t & 2u) << 15);
        w3 = ((t >> 2) & 1u) | ((t & 8u) << 13);
    };

    if (wave >= 8) {
        // ======================================================================================== gather waves
        constexpr int NGW = 4;
        if (KRK_DBGBIT(a, 32)) __builtin_amdgcn_s_setprio(3);   // probe bit 32: the gather waves ahead of the compute waves of their SIMD (slower: +7 %)
        constexpr int NGP = (NPEER * 8 * 16 + 64 * NGW - 1) / (64 * NGW);   // granules per gather lane (a slice publishes <= 8 blocks x 16 lines)
        static_assert(NGP == 2 || NGP == 4, "the vmcnt(0) asm below lists NGP registers");
        const int gl = tid - 512;                                 // 0 .. 64*NGW-1
        unsigned g_vo[NGP], g_lds[NGP];
        bool g_on[NGP];
        const unsigned dump_off = dump_base + (unsigned)tid * 16u;
#pragma unroll
        for (int j = 0; j < NGP; ++j) {
            const unsigned q = (unsigned)gl + (unsigned)(64 * NGW) * j;
            const unsigned p = q / slice_gran, rem = q - p * slice_gran;
            const int sl = (slice + 1 + (int)p) % CS;
            const int blk = (int)(rem >> 4), ln = (int)(rem & 15);
            const int unit = (sl * BPC + blk) * 4;
            g_on[j] = p < (unsigned)NPEER && (sl * BPC + blk) < a.NB;          // blocks beyond NB are never published
            g_vo[j] = g_on[j] ? ((unsigned)sl * slice_gran + (unsigned)(blk * 16 + ln)) * 16u : kOOBwp;
            g_lds[j] = (g_on[j] && unit < NKB * 32) ? lds_of(ln, unit) : 0xFFFFFFFFu;
        }
        bool dead = false;
        u32x4 gd[NGP];
        auto issue = [&](int tg, int ts) {                        // granules of h(tg, ts - 1)
            const unsigned so = (unsigned)(tg * 2 + (ts & 1)) * gp_bytes;
#pragma unroll
            for (int j = 0; j < NGP; ++j) wp_load_b128_sc1(gd[j], g_vo[j], grs, so);
        };
        auto land = [&]() {
            if constexpr (NGP == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(gd[0]), "+v"(gd[1]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(gd[0]), "+v"(gd[1]), "+v"(gd[2]), "+v"(gd[3]) : : "memory");
        };
        // h(tg, ts - 1) -> gd.  `inflight`: the request went out at the end of the previous iteration (after that iteration's rows
        // were written: the registers are free then, and nothing is in flight across the loop's back edge that the compiler could copy)
        unsigned gkk = 0;
        auto gather = [&](int tg, int ts, bool inflight) {
            unsigned w2, w3;
            tagbits(ts, w2, w3);
            unsigned spins = 0;
            while (true) {
                if (!inflight) issue(tg, ts);
                inflight = false;
                land();
                bool ok = true;
#pragma unroll
                for (int j = 0; j < NGP; ++j) ok = ok && (!g_on[j] || ((gd[j][2] & 0x00010001u) == w2 && (gd[j][3] & 0x00010001u) == w3));
                if (spins == 0) WP_STAMP(wave == 8, gkk, 5);
                if (__all(ok) || dead) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) {          // ~ a second: give up, flag the plan, never wait again
                    dead = true;
                    // what was missing, for the post-mortem: step (12 bits) | group << 12 | slice << 14 | the tag bits seen instead << 20
                    unsigned seen = 0;
#pragma unroll
                    for (int j = 0; j < NGP; ++j)
                        if (g_on[j] && ((gd[j][2] & 0x00010001u) != w2 || (gd[j][3] & 0x00010001u) != w3))
                            seen = (gd[j][2] & 1u) | ((gd[j][2] >> 15) & 2u) | ((gd[j][3] & 1u) << 2) | ((gd[j][3] >> 13) & 8u);
                    const unsigned long long bad = __ballot(!ok);
                    const int first = bad ? __builtin_ctzll(bad) : 0;
                    seen = (unsigned)__builtin_amdgcn_readlane((int)seen, first);
                    if (lane == 0) __hip_atomic_store(a.err, 0x80000000u | ((unsigned)ts & 0xFFFu) | ((unsigned)tg << 12) | ((unsigned)slice << 14) | ((seen & 0x7FFu) << 20),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            WP_STAMP(wave == 8, gkk, 6);
        };
        auto rows = [&](int tg) {                                 // gd -> the rows of LDS buffer tg
            unsigned char* hb = hs + tg * hbuf;
#pragma unroll
            for (int j = 0; j < NGP; ++j) {
                const bool nowhere = g_lds[j] == 0xFFFFFFFFu;
                unsigned char* dst = nowhere ? smem8 + dump_off : hb + g_lds[j];
                u32x2 hi2, lo2;
                hi2[0] = gd[j][0]; hi2[1] = gd[j][1];
                lo2[0] = gd[j][2] & 0xFFFEFFFEu; lo2[1] = gd[j][3] & 0xFFFEFFFEu;    // the tag bits are not part of the value
                *reinterpret_cast<u32x2*>(dst) = hi2;
                *reinterpret_cast<u32x2*>(dst + (nowhere ? 8 : plane)) = lo2;
            }
        };
        // Iteration (s, g) runs beside compute stage (g, s) and prepares the input of the NEXT stage: (g+1, s) consumes h(g+1, s-1)
        // -- or (0, s+1) consumes h(0, s); in the epilogue (s == Lmax) the "stages" only write h(., Lmax-1) out.
        // Tried (round 3, profiles/r03_lstm_wp_gather_variants.txt): landing registers the compiler does not own (kernel capped with
        // amdgpu_num_vgpr, the asm naming v152.. directly), so that the request for the next target can go out BEFORE the rows are
        // written and stay in flight across the loop's back edge -- correct, and no faster: the stage is not bound by the exchange
        // (see DESIGN.md 3.3), and the register copies cost 6 %.
        auto target = [&](int s_, int g_, int& tg, int& ts) -> bool {       // what iteration (s_, g_) gathers; false: nothing
            if (g_ + 1 < NG) { tg = g_ + 1; ts = s_; return s_ > 0; }
            tg = 0; ts = s_ + 1;
            return s_ < Lmax;
        };
        bool pend = false;                                        // the request of this iteration's target is in flight
        for (int s = 0; s <= Lmax; ++s) {
#pragma unroll
            for (int g = 0; g < NG; ++g, ++gkk) {
                WP_STAMP(wave == 8, gkk, 7);
                WP_STAMP(true, gkk, 8 + wave);
                wp_barrier();
                WP_STAMP(true, gkk, 20 + wave);
                WP_STAMP(wave == 8, gkk, 4);
                int tg, ts, ng, ns;
                const bool have = target(s, g, tg, ts);
                if (have) {
                    gather(tg, ts, pend);
                    rows(tg);
                }
                // the target of the next iteration was published during the stage before this one: ask for it now (not across the
                // loop's back edge: the compiler may copy a register an unfinished load will write)
                const bool more = g + 1 < NG && target(s, g + 1, ng, ns);
                if (more) issue(ng, ns);
                pend = more;
            }
        }
        WP_STAMPS_OUT();
        return;
    }

    // ============================================================================================ compute waves
    const int line = lane & 15, us = lane >> 4;
    const unsigned dump_off = dump_base + (unsigned)tid * 16u;
    // this wave's block: local block `wave` of the slice, global block slice*BPC + wave
    const bool bval = (wave < BPC) && (slice * BPC + wave < a.NB);          // wave-uniform

    // ---- weights: resident for the whole launch.  [dir][slice CS][wave 8][kb][plane][lane][8]
    u32x4 whi[NKB], wlo[NKB];
    {
        const __bf16* wb = a.wp + ((((size_t)dir * CS + slice) * 8 + wave) * NKB) * 1024 + lane * 8;
        // pins these loads BELOW the role branch: hoisted above it (they are speculatable) they cost the gather waves 8 NKB
        // registers they have no room for
        asm volatile("; weights of the compute waves" : "+v"(wb));
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            whi[kb] = *reinterpret_cast<const u32x4*>(wb + (size_t)kb * 1024);
            wlo[kb] = *reinterpret_cast<const u32x4*>(wb + (size_t)kb * 1024 + 512);
        }
    }

    // ---- xproj: rows are tile-time-major, this cluster's NG 16-line groups are NG consecutive tiles of T*16 rows.
    // HBM -> LDS directly (lanes without a valid row -- finished lines, an absent block, a group past N -- read row 0 of the
    // cluster's first tile; whatever they compute stays in their own MFMA column / dump word).
    const int ntiles = min((a.N - n0 + 15) / 16, NG);
    const float* xbase = a.xp + (size_t)n0 * a.T * a.xstride + (size_t)dir * a.G + (size_t)line * a.xstride + us * 4;
    const unsigned xcol = bval ? (unsigned)(slice * BPC + wave) * 16u : 0u;
    int len_x[NG];                                                // length of this lane's line in every group (registers: no LDS round trip per stage)
    auto line_len = [&](int n) -> int { return n < a.N ? (a.lens ? min(max(a.lens[n], 0), a.T) : a.T) : 0; };
#pragma unroll
    for (int g = 0; g < NG; ++g) len_x[g] = (g < ntiles) ? line_len(n0 + 16 * g + line) : 0;
    // the address of a group's NEXT xproj row is carried, not recomputed: one 64-bit add per load instead of two 64-bit
    // multiplies (the stage is issue-bound: ~190 instructions per compute wave, 3 waves per SIMD)
    const float* xdummy = xbase + xcol;                           // row 0 of the cluster's first tile: what lanes without a row read
    const long long xstep = (long long)(rev ? -16 : 16) * a.xstride;          // floats per time step of a line
    const float* xnext[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int t0 = rev ? (len_x[g] - 1) : 0;
        xnext[g] = xbase + ((size_t)g * a.T + (size_t)max(t0, 0)) * 16 * a.xstride + xcol;
    }
    auto load_x = [&](int g, int s, unsigned ring) {              // must be called for s = 0, 1, 2, ... of a group, once each
        const float* src = (s < len_x[g] && !KRK_DBGBIT(a, 8)) ? xnext[g] : xdummy;    // probe bit 8: no HBM stream (one hot row)
        xnext[g] += xstep;
        if (KRK_DBGBIT(a, 64)) return;                            // probe bit 64: no xproj copy at all (the ring holds zeros)
        wp_load_lds_b128(src, xs_off + (ring * 8u + (unsigned)wave) * 1024u);
    };

    // ---- what this wave publishes: its block = units slice*BPC*4 + wave*4 + (0..3).  Lane (line, us) computes unit us of its
    // line; the four units of a line are collected into the lanes us == 0 (publish() below), which write the LDS rows and the granule
    const int unit0 = (slice * BPC + wave) * 4;
    const bool pubber = bval && us == 0;
    const unsigned pub_vo = pubber ? ((unsigned)slice * slice_gran + (unsigned)(wave * 16 + line)) * 16u : kOOBwp;
    const unsigned own_lds = (pubber && unit0 < NKB * 32) ? lds_of(line, unit0) : 0xFFFFFFFFu;

    // ---- output pass, one 16-byte piece per lane (tid < 512).  Tile-time-major rows (a.otiled: the consumer is gemm_x3, which
    // keeps the order): the 16 lines of a group at one step are 16 consecutive rows, a slice takes every CS-th (plane, piece)
    // combination for ALL 16 lines -- whole 256-byte runs.  Line-major rows: the slice writes 16/CS lines of a group.
    const int per_line = a.H >> 3;
    const size_t rows_total = a.otiled ? (size_t)((a.N + 15) / 16 * 16) * a.T : (size_t)a.N * a.T;
    const i32x4 ors = wp_srd(a.out, (unsigned)((size_t)a.out_plane * 4));
    unsigned sp_lds, sp_g00, sp_tmul;
    int sp_ln;
    if (a.otiled) {
        const int cmb = slice + CS * (tid >> 4);                // (plane, piece) combination, plane-major
        const int pl = cmb / per_line, q = cmb - pl * per_line;
        sp_ln = cmb < 2 * per_line ? (tid & 15) : -1;
        sp_lds = sp_ln >= 0 ? (unsigned)(pl * plane + (q & 3) * OS + sp_ln * RSO + (q >> 2) * 16) : 0u;
        sp_g00 = (unsigned)((((size_t)(dir * per_line + q)) * rows_total + (size_t)(n0 >> 4) * a.T * 16 + (size_t)max(sp_ln, 0)) * 16 + (size_t)pl * a.out_plane * 2);
        sp_tmul = 16u * 16u;                                      // bytes per time step: 16 rows
    } else {
        constexpr int LPS = 16 / CS;                              // lines of a group this slice writes
        const int e = tid;
        const int pl = e / (LPS * per_line), r = e - pl * LPS * per_line;
        const int li = r / per_line, q = r - li * per_line;
        sp_ln = e < 2 * LPS * per_line ? slice * LPS + li : -1;
        sp_lds = sp_ln >= 0 ? (unsigned)(pl * plane + (q & 3) * OS + sp_ln * RSO + (q >> 2) * 16) : 0u;   // piece q = units 8q..8q+7
        sp_g00 = (unsigned)((((size_t)(dir * per_line + q)) * rows_total + (size_t)(n0 + max(sp_ln, 0)) * a.T) * 16 + (size_t)pl * a.out_plane * 2);
        sp_tmul = 16u;                                            // consecutive steps of a line are consecutive rows
    }
    const unsigned sp_gmul = 16u * (unsigned)a.T * 16u;           // next group: 16*T rows further in both orders
    // the store offset of a group's NEXT output step is carried the same way
    int len_sp[NG];
    unsigned vnext[NG];
    const unsigned vstep = rev ? 0u - sp_tmul : sp_tmul;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        len_sp[g] = sp_ln >= 0 ? line_len(n0 + 16 * g + sp_ln) : 0;
        vnext[g] = sp_g00 + (unsigned)g * sp_gmul + (unsigned)(rev ? max(len_sp[g] - 1, 0) : 0) * sp_tmul;
    }
    auto store_read = [&](int g, int step, const unsigned char* hb, unsigned& vo) -> u32x4 {       // step = -1, 0, 1, ... of a group, once each
        const bool on = step >= 0 && step < len_sp[g];
        vo = (on && !KRK_DBGBIT(a, 16)) ? vnext[g] : kOOBwp;                       // probe bit 16: no output stores
        if (step >= 0) vnext[g] += vstep;
        return *reinterpret_cast<const u32x4*>(hb + sp_lds);
    };

    float cst[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) cst[g] = 0.f;
    f32x4 zq = f32x4{0.f, 0.f, 0.f, 0.f};        // pre-activations of the PREVIOUS stage (its gates run inside this one)

    // The gates of stage (pg, ps) -- krk_lstm_cell (common.h) cut into three pieces that run BEHIND the MFMAs of the first three K
    // blocks of the next stage (in-order issue: a wave's VALU work delays its own later MFMAs, not the ones already in the matrix
    // pipe, and the other compute wave of the SIMD fills the gaps).  Left to the compiler the whole chain -- 7 transcendentals, ~35
    // VALU -- was scheduled in FRONT of the stage's first LDS read (profiles/r03_lstm_wp_gather_variants.txt: the first MFMA of a
    // stage issued ~700 cycles after the barrier).  c, h split into bf16 hi + a lo part of 7 significant bits (its lowest mantissa
    // bit carries the granule's tag), packed hi | lo << 16.
    struct GateRegs { float ei, ef, eg, eo, po, et; };
    auto gates_a = [&](GateRegs& q) {                              // the four exponentials of the pre-activations
        constexpr float L2E = 1.4426950408889634f, LIM = 28.853900817779268f;   // 20 log2 e
        q.ei = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-L2E * zq[0], -LIM, LIM));
        q.ef = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-L2E * zq[1], -LIM, LIM));
        q.eg = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-2.f * L2E * zq[2], -LIM, LIM));
        q.eo = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-L2E * zq[3], -LIM, LIM));
    };
    auto gates_b = [&](GateRegs& q, int pg) {                      // c' = f c + i tanh g over one denominator; e^-2c'
        constexpr float L2E = 1.4426950408889634f, LIM = 28.853900817779268f;
        const float pi = 1.f + q.ei, pf = 1.f + q.ef, pgg = 1.f + q.eg;
        q.po = 1.f + q.eo;
        const float pig = pi * pgg;
        const float num = __builtin_fmaf(cst[pg], pig, (1.f - q.eg) * pf);
        const float cn = num * __builtin_amdgcn_rcpf(pig * pf);
        cst[pg] = cn;
        q.et = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-2.f * L2E * cn, -LIM, LIM));
    };
    auto gates_c = [&](const GateRegs& q) -> unsigned {            // h = o tanh c', split
        const float h = (1.f - q.et) * __builtin_amdgcn_rcpf(q.po * (1.f + q.et));
        const __bf16 hb16 = (__bf16)h;
        const unsigned lf = __builtin_bit_cast(unsigned, h - (float)hb16);
        const unsigned lr = (lf + 0xFFFFu + ((lf >> 17) & 1u)) & 0xFFFE0000u;        // round to nearest even at bit 17
        return (unsigned)__builtin_bit_cast(unsigned short, hb16) | lr;
    };
    auto gates = [&](int pg) -> unsigned {                         // all three at once (the epilogue)
        GateRegs q;
        gates_a(q);
        gates_b(q, pg);
        return gates_c(q);
    };
    // ... and where they go.  The four units of a line sit in the lanes line + 16 us: two row swaps bring all four to every lane
    // (v_permlane16_swap: odd rows of the first operand <-> even rows of the second; v_permlane32_swap: rows 2, 3 of the first <->
    // rows 0, 1 of the second); the lanes us == 0 write the 8 + 8 bytes of the (line, block) position of h(pg, ps) into LDS
    // buffer pg and the same 16 bytes, tagged ps + 1, into the granule.
    auto publish = [&](unsigned P, int pg, int ps, bool real) {
        const auto r = __builtin_amdgcn_permlane16_swap(P, P, false, false);          // r[0] rows: P0 P0 P2 P2, r[1] rows: P1 P1 P3 P3
        const auto e = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false);    // e[0] = P0 everywhere, e[1] = P2
        const auto o = __builtin_amdgcn_permlane32_swap(r[1], r[1], false, false);    // o[0] = P1 everywhere, o[1] = P3
        unsigned w2, w3;
        tagbits(ps + 1, w2, w3);
        u32x4 gran;
        gran[0] = (e[0] & 0xFFFFu) | (o[0] << 16);
        gran[1] = (e[1] & 0xFFFFu) | (o[1] << 16);
        gran[2] = (e[0] >> 16) | (o[0] & 0xFFFF0000u);
        gran[3] = (e[1] >> 16) | (o[1] & 0xFFFF0000u);
        const bool nowhere = own_lds == 0xFFFFFFFFu;
        unsigned char* dst = nowhere ? smem8 + dump_off : hs + pg * hbuf + own_lds;
        u32x2 hi2, lo2;
        hi2[0] = gran[0]; hi2[1] = gran[1];
        lo2[0] = gran[2]; lo2[1] = gran[3];
        *reinterpret_cast<u32x2*>(dst) = hi2;
        *reinterpret_cast<u32x2*>(dst + (nowhere ? 8 : plane)) = lo2;
        gran[2] |= w2;
        gran[3] |= w3;
        wp_store_b128_so(gran, real ? pub_vo : kOOBwp, grs, (unsigned)(pg * 2 + ((ps + 1) & 1)) * gp_bytes);
    };

    // The weight loads above are the compiler's own: make it wait for them HERE, with an instruction its wait-count pass
    // sees -- otherwise it waits at their first use inside the time loop, `s_waitcnt vmcnt(0)` in the middle of every stage,
    // which also waits for the hand-issued stores it does not know about.
    __builtin_amdgcn_s_waitcnt(0);
    // ---- prologue: xproj of stages 0, 1, 2 (the fixed-count wait of a stage assumes three full stages behind it)
    load_x(0 % NG, 0 / NG, 0u);
    load_x(1 % NG, 1 / NG, 1u);
    load_x(2 % NG, 2 / NG, 2u);
    wp_vmwait<0>();
    unsigned kk = 0;                                               // stage counter
    for (int s = 0; s < Lmax; ++s) {
#pragma unroll
        for (int g = 0; g < NG; ++g, ++kk) {
            // ONE h buffer per group is enough because the gates are deferred: buffer g holds h(g, s-1) while stage (g, s) reads it
            // (fragments, output piece); our own rows of h(g, s) are written during the NEXT stage and the peers' rows during
            // the stage before (g, s+1) -- both behind the barrier that ends this stage's reads.
            const unsigned char* hb = hs + g * hbuf;                // h(g, s-1): own rows written by our gates, the rest gathered
            const int pg = (g + NG - 1) % NG, ps = g == 0 ? s - 1 : s;
            WP_STAMP(wave == 0, kk, 3);
            WP_STAMP(wave == 0 || !KRK_DBGBIT(a, 8192), kk, 8 + wave);
            wp_barrier();
            WP_STAMP(true, kk, 20 + wave);
            WP_STAMP(wave == 0, kk, 0);
            // per stage a wave with a block issues exactly [xproj load, output store, publish store]; xproj of THIS stage was the
            // first of the three issued three stages ago: 2 + 3 + 3 younger operations may still be in flight.  (A wave without
            // a block only stores: nothing to wait for.)
            if (KRK_DBGBIT(a, 512)) { if (wave >= 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
            if (KRK_DBGBIT(a, 1024)) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }           // probe: the younger wave of a SIMD first, all stage long
            if (bval) { if (KRK_DBGBIT(a, 1)) wp_vmwait<0>(); else wp_vmwait<8>(); }
            const unsigned ring = kk & (RING - 1);
            const int g3 = (g + 3) % NG, s3 = s + (g + 3) / NG;     // xproj of stage kk + 3
            if (bval) {
                // LDS returns in order: this stage's operands first (xproj landing, the first two K blocks of h), then the
                // output piece of the previous step
                f32x4 acc0 = *reinterpret_cast<const f32x4*>(smem8 + xs_off + (ring * 8u + (unsigned)wave) * 1024u + lane * 16);
                if (KRK_DBGBIT(a, 2)) {                            // probe: xproj straight from memory, no landing ring
                    const int len = len_x[g];
                    const int t = rev ? (len - 1 - s) : s;
                    const size_t row = (s < len && g < ntiles) ? ((size_t)g * a.T + t) * 16 : 0;
                    acc0 = *reinterpret_cast<const f32x4*>(xbase + row * a.xstride + xcol);
                    __builtin_amdgcn_s_waitcnt(0);
                }
                f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
                // fragments two K blocks ahead of their MFMAs: three blocks live (24 registers), not all NKB (8 NKB)
                bf16x8 hh[NKB], hl[NKB];
                auto frag = [&](int kb) {
                    const unsigned char* hp = hb + us * OS + line * RSO + kb * 16;
                    if (KRK_DBGBIT(a, 2048) && kb > 1) { hh[kb] = hh[kb & 1]; hl[kb] = hl[kb & 1]; return; }   // probe: two fragment reads instead of NKB
                    hh[kb] = *reinterpret_cast<const bf16x8*>(hp);
                    hl[kb] = *reinterpret_cast<const bf16x8*>(hp + plane);
                };
                frag(0);
                if (NKB > 1) frag(1);
                unsigned sp_vo;
                const u32x4 sp_v = store_read(g, s - 1, hb, sp_vo);
                __builtin_amdgcn_sched_barrier(0);                  // the stage's first LDS reads go out before anything else
                WP_STAMP(wave == 0, kk, 1);
                // stage (pg, ps)'s gate math in three pieces behind the MFMAs of K blocks 0, 1, 2 (fewer blocks: the last takes
                // what is left); its LDS rows and its granule leave with block PUBK
                constexpr int PUBK = NKB > 4 ? 3 : (NKB - 1);
                constexpr int KA = 0, KB = NKB > 1 ? 1 : 0, KC = NKB > 2 ? 2 : NKB - 1;
                GateRegs gq;
                unsigned go = 0;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    if (kb + 2 < NKB) frag(kb + 2);
                    if (KRK_DBGBIT(a, 512) && kb == (NKB - 1) / 2) { if (wave >= 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }   // probe: the two compute waves of a SIMD swap priority mid-stage
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_bf(whi[kb]), hh[kb], acc0, 0, 0, 0);
                    KRK_CROSS(acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_bf(whi[kb]), hl[kb], acc1, 0, 0, 0);
                              acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_bf(wlo[kb]), hh[kb], acc2, 0, 0, 0);)
                    __builtin_amdgcn_sched_barrier(0);              // the block's MFMAs are in the pipe before its share of VALU work issues
                    if (kb == KA) gates_a(gq);
                    if (kb == KB) gates_b(gq, pg);
                    if (kb == KC) go = gates_c(gq);
                    if (kb == 0) load_x(g3, s3, (kk + 3u) & (RING - 1));           // vector memory in the order [xproj, output, publish]
                    if (kb == (NKB > 1 ? 1 : 0)) wp_store_b128(sp_v, sp_vo, ors);
                    if (kb == PUBK) publish(go, pg, ps, kk != 0);
                    __builtin_amdgcn_sched_barrier(0);
                    WP_STAMP(wave == 0 && KRK_DBGBIT(a, 8192) && acc0[0] != 123.f, kk, 9 + kb);   // probe: wave 0's K loop, block by block
                }
                zq = acc0 + (acc1 + acc2);
                WP_STAMP(wave == 0 && zq[0] != 123.f, kk, 2);
            } else {                                                // a wave without a block: its share of the output pass only
                (void)g3; (void)s3;
                unsigned sp_vo;
                const u32x4 sp_v = store_read(g, s - 1, hb, sp_vo);
                wp_store_b128(sp_v, sp_vo, ors);
            }
        }
    }
    // ---- epilogue: the gates of the last stage, then h(g, Lmax-1) of every group leaves (the gather waves complete it)
    {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const unsigned char* hb = hs + g * hbuf;
            wp_barrier();
            if (g == 0 && bval) publish(gates(NG - 1), NG - 1, Lmax - 1, true);
            unsigned sp_vo;
            const u32x4 sp_v = store_read(g, Lmax - 1, hb, sp_vo);
            wp_store_b128(sp_v, sp_vo, ors);
        }
    }
    WP_STAMPS_OUT();
}

template <int NKB, int NG, int CS>
int launch_wp(const LstmWsArgs& a, hipStream_t s) {
    const int nclusters = (a.N + 16 * NG - 1) / (16 * NG) * a.ndir;
    const size_t lds = (size_t)NG * 2 * 4 * 16 * 16 * (NKB | 1) + 16 * NG * sizeof(int) + 16 + 768 * 16 + (size_t)4 * 8 * 1024 + 8192 /* timeline stamps of the ablation build */;
    auto kfn = lstm_wp_kernel<NKB, NG, CS>;
    if (lds > 160 * 1024) return -4;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (nclusters != a.nclusters) return -1;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(nclusters * CS + 8 * (CS - 1))), dim3(768), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

#ifndef KRK_BF16_ONE
// slices per cluster for a hidden size: one gate-column block per compute wave, 8 compute waves per slice
int krk_lstm_wp_slices(int Hp) { return Hp / 4 <= 32 ? 4 : 8; }

bool krk_lstm_wp_supported(int H, int Hp) {
    const int NKB = (Hp + 31) / 32, NB = Hp / 4;
    return (H % 8) == 0 && NKB >= 1 && NKB <= 7 && NB <= 8 * krk_lstm_wp_slices(Hp);
}

int krk_lstm_wp_clusters(int N, int ndir) { return (N + 63) / 64 * ndir; }

// ctrl block of a launch (zeroed by the host before every launch): [0..7] arrivals per XCD, [8] work items claimed, [9] posted,
// [16 + x * mbox + i] mailbox of the i-th local group of XCD x; mbox = groups one XCD can see if EVERY block lands on it
int krk_lstm_wp_mbox(int nclusters, int Hp) { const int CS = krk_lstm_wp_slices(Hp); return nclusters + 8 * (CS - 1) / CS + 1; }
size_t krk_lstm_wp_ctrl_bytes(int nclusters, int Hp) { return (size_t)(16 + 8 * krk_lstm_wp_mbox(nclusters, Hp)) * 4; }

size_t krk_lstm_wp_gran_bytes(int N, int ndir, int Hp) {
    const int CS = krk_lstm_wp_slices(Hp), BPC = (Hp / 4 + CS - 1) / CS;
    return (size_t)krk_lstm_wp_clusters(N, ndir) * 8 /* (group, parity) */ * CS * (size_t)BPC * 16 * 16;
}
#endif

// a.BPC = ceil(NB / CS) blocks per slice, a.wp = [dir][slice CS][wave 8][kb][plane][lane][8]
int KRK_FN(krk_launch_lstm_wp)(const LstmWsArgs& a, hipStream_t s) {
    if (!krk_lstm_wp_supported(a.H, a.Hp)) return -4;
    if ((size_t)a.out_plane * 4 >= 0x80000000ull) return -4;                 // 32-bit buffer offsets
    const int CS = krk_lstm_wp_slices(a.Hp);
    if ((size_t)8 * CS * a.BPC * 16 * 16 >= 0x80000000ull) return -4;
#define KRK_WP(NKB_, CS_) if (a.NKB == NKB_ && CS == CS_) return launch_wp<NKB_, 4, CS_>(a, s)
    KRK_WP(1, 4); KRK_WP(2, 4); KRK_WP(3, 4); KRK_WP(4, 4); KRK_WP(5, 8); KRK_WP(6, 8); KRK_WP(7, 8);
#undef KRK_WP
    return -4;
}
