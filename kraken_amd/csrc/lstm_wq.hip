// Weight-stationary recurrent kernel of the (bi)directional LSTM, second generation: ONE wave per SIMD, W_hh in the ACCUMULATOR
// half of the register file.  Reference semantics: nn.LSTM inside TransposedSummarizingRNN.forward
// (kraken/lib/vgsl/layers.py:513-547): packed by length, gates i,f,g,o, h/c start at zero, outputs past a line's length stay zero.
//
// lstm_ws.hip (rounds 2-4) ran a cluster slice as 8 waves x <= 256 registers, two per SIMD in lockstep behind one workgroup barrier
// per slot: per slot and SIMD 1340 MFMA cycles, but also 2 x 187 VALU instructions (every wave pays the gather / address / output
// bookkeeping) and 8 x 14 KB of h fragments out of LDS -- phases that ran one after the other (3600 cycles per slot, matrix pipe
// 37 % busy; without any MFMA the slot still took 3000).  Here
//   * a slice is 4 waves (one per SIMD) with the whole 512-entry register file each: the wave's <= 4 gate-column blocks x NKB
//     K blocks x (hi, lo) fragments live in AGPRs (224 at H = 200) and feed v_mfma_f32_16x16x32_bf16 DIRECTLY as its A operand
//     (inline asm, constraint "a": hipcc would copy them through VGPRs); the 256 architectural VGPRs hold everything else, so
//     every offset of the time loop is precomputed instead of recomputed;
//   * a wave reads the h fragments ONCE for four blocks (half the LDS bytes per MFMA of lstm_ws) and pays the per-wave
//     bookkeeping once for twice the matrix work;
//   * a granule PAIR (two adjacent units of one line) is 16 contiguous bytes {payload u, tag, payload u+1, tag}: one
//     buffer_load_dwordx4 per pair on the gathering side (same tags, same protocol as lstm_ws: each 8-byte half is validated by its
//     own tag);
//   * cluster = 4 workgroups (one per CU), two 16-line groups in alternation, data-tagged sc1 exchange, run-time membership,
//     bounded spins, heartbeat granules of block-less slices, tile-time-major xproj / output rows: as in lstm_ws.hip (see there).
// Arithmetic is that of lstm_ws.hip instruction for instruction (three accumulators per block, z = a0 + (a1 + a2), krk_lstm_cell):
// the two kernels agree bit for bit.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef KRK_STAMP
#define WQ_STAMP(i) do { if (a.stamps) { const unsigned long long n_ = __builtin_readcyclecounter(); st_acc[i] += n_ - st_t; st_t = n_; } } while (0)
#define WQ_COUNT(i) do { if (a.stamps) st_acc[i] += 1; } while (0)
#else
#define WQ_STAMP(i) do { } while (0)
#define WQ_COUNT(i) do { } while (0)
#endif

namespace {

constexpr unsigned kOOBwq = 0x80000000u;   // voffset beyond every descriptor used here (all < 2 GiB): load = 0, store dropped

typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 wq_srd(const void* p, unsigned bytes) {       // raw buffer descriptor in SGPRs
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    i32x4 r;
    r[0] = (int)__builtin_amdgcn_readfirstlane((unsigned)u);
    r[1] = (int)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xFFFFu);
    r[2] = (int)__builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}
// ---- vector memory by hand (lstm_ws.hip: vmcnt retires in order, an sc1 store takes ~1 us to be acknowledged, so every wait is counted)
__device__ __forceinline__ void wq_load_b128_sc1(u32x4& d, unsigned vo, const i32x4& srd, unsigned so) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "=&v"(d) : "v"(vo), "s"(srd), "s"(so) : "memory");
}
// publish: a PLAIN store.  A cluster lives on ONE XCD (see the cluster claim), whose L2 is the point of coherence of its CUs: the
// store writes through the CU's L1 into that L2 and stays there; the peers' polls (sc1 loads: bypass L1, served by L2) hit it a few
// hundred cycles later.  (An sc1 store writes through to the fabric: ~1.2 us until a poll from another XCD returns it.)
__device__ __forceinline__ void wq_store_b64(const u32x2& d, unsigned vo, const i32x4& srd, unsigned so) {
    asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" : : "v"(d), "v"(vo), "s"(srd), "s"(so) : "memory");
}
__device__ __forceinline__ void wq_store_b128(const u32x4& d, unsigned vo, const i32x4& srd) {
    // s_nop: wait states of the ">64-bit store data, then VALU write of those registers" hazard (invisible to the compiler here)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(d), "v"(vo), "s"(srd) : "memory");
}
__device__ __forceinline__ void wq_load_lds_b128(const float* gptr, unsigned lds_off) {   // 64 lanes x 16 B -> LDS [lds_off, +1 KB)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_off) : "memory");
}
template <int N>
__device__ __forceinline__ void wq_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// The resident weight fragments are pinned to the accumulator file by empty asm statements (constraint "+a") in front of their
// uses: v_mfma takes its A operand from an AGPR as it stands, so the builtin -- which hipcc schedules, pads and interleaves
// like any MFMA -- reads the fragment where it lies, and no register allocation decision can pull 224 registers of weights into
// the 256 architectural VGPRs.
__device__ __forceinline__ void wq_pin(u32x4& hi, u32x4& lo) { asm("" : "+a"(hi), "+a"(lo)); }   // not volatile: free to move with its MFMAs
__device__ __forceinline__ bf16x8 wq_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// NKB K blocks of 32, NW waves per workgroup (4: one per SIMD, 512 registers each; 8: two per SIMD, 256 each), BPW gate-column
// blocks per wave (local block = wave + NW i), two 16-line groups per cluster
template <int NKB, int BPW, int NW>
__global__ void __launch_bounds__(64 * NW) lstm_wq_kernel(const LstmWsArgs a) {
    constexpr int NT = 64 * NW;
    constexpr int NG = 2;
    // a lane gathers NGP granule PAIRS per (group, step): 3 peers x BPC*32 pairs / NT lanes, BPC <= NW BPW
    constexpr int NGP = (3 * BPW + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    // h in LDS as in lstm_ws.hip, per (group, parity): [plane hi|lo][K octet 4][line 16][K block: 16 bytes each, odd count]
    constexpr int RSO = 16 * (NKB | 1);
    constexpr int OS = 16 * RSO;
    constexpr int plane = 4 * OS;
    // every plane is followed by a dump strip (one dword per thread): masked writes -- absent blocks, padding units, heartbeat
    // pairs -- go there through the SAME address arithmetic as real ones (lo = hi + planeP), no select in the time loop
    constexpr int planeP = plane + 4 * NT;
    constexpr int hbuf = 2 * planeP;
    auto lds_of = [&](int ln, int unit) -> unsigned {        // byte offset of (line, unit) inside a buffer's hi plane
        return (unsigned)(((unit & 31) >> 3) * OS + ln * RSO + (unit >> 5) * 16 + (unit & 7) * 2);
    };
    unsigned char* hs = smem8;              // [group NG][parity 2][hbuf]
    int* lens_s = reinterpret_cast<int*>(smem8 + 2 * NG * hbuf);        // [16 * NG]
    unsigned* misc = reinterpret_cast<unsigned*>(lens_s + 16 * NG);     // [0] ticket
    const unsigned xs_off = (unsigned)(2 * NG * hbuf + 16 * NG * 4 + 16 + 256) & ~255u;   // xproj landing ring [slot parity 2][wave NW][BPW][64 lanes x 16 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- cluster claim: FOUR workgroups that run on the SAME XCD (its L2 is then the point of coherence of the exchange: plain
    // stores, polls served by L2 -- a hop costs ~780 cycles instead of ~1750 across XCDs, tools/ubench/handoff_probe, and nothing is
    // written through to HBM).  Arrivals are counted per XCD (ctrl[x]); the workgroup that completes a local group of four claims a
    // work item c (ctrl[8]) and posts it in the group's mailbox; the others wait for the mailbox.  The grid holds 8 x 3 workgroups
    // more than the C clusters need, so that C full groups form whatever the distribution of blocks over XCDs
    // (sum_x floor(n_x / 4) >= (G - 24) / 4 = C); with the round-robin placement observed (block b on XCD b % 8) the surplus blocks
    // find all work claimed and leave at once.  Progress needs only four co-resident workgroups on one XCD -- a partially resident
    // grid cannot deadlock -- and a group that can never fill ends when all C items are posted.
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned C = (unsigned)a.nclusters;
        const unsigned t = atomicAdd(a.ctrl + xcc, 1u);
        const unsigned lc = t >> 2, sl = t & 3u;
        unsigned* mb = a.ctrl + 16 + xcc * (unsigned)a.mbox + lc;
        unsigned c = C;
        if (lc < (unsigned)a.mbox) {
            if (sl == 3u) {
                c = atomicAdd(a.ctrl + 8, 1u);
                __hip_atomic_store(mb, c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ctrl[9] counts the POSTED work items: a waiting workgroup may only conclude "my group is surplus" from a counter
                // that is bumped after the mailboxes are visible (ctrl[8] is bumped before: a member of the group that took the
                // last item would see "all claimed" ahead of its own mailbox and leave its cluster short)
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
    for (int e = tid; e < NG * hbuf / 2; e += NT) reinterpret_cast<unsigned int*>(hs)[e] = 0u;   // 2*NG*hbuf bytes
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

    const int line = lane & 15, us = lane >> 4;
    const unsigned dump_off = (unsigned)plane + (unsigned)tid * 4u;      // inside every h buffer: the strip behind the hi plane
    const int BPC = a.BPC;

    // which of this wave's blocks exist (wave-uniform): local block wave + NW i < BPC, global block < NB
    bool bval[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) bval[i] = (wave + NW * i < BPC) && (slice * BPC + wave + NW * i < a.NB);

    // ---- weights: resident in AGPRs for the whole launch, straight from the streaming kernel's layout
    // [dir][kb][block][plane][lane][8] (capi.hip: upload_lstm_x3).  The loads WRITE accumulator registers (constraint "=a"): a
    // value that passes through hipcc's hands as a VGPR is copied into one scratch AGPR quad in front of every MFMA instead of
    // staying put.  Blocks that do not exist read beyond the descriptor: zeros.
    u32x4 whi[BPW][NKB], wlo[BPW][NKB];
    {
        const unsigned kb_bytes = (unsigned)a.NB * 2048u;
        const i32x4 wrs = wq_srd(a.wp + (size_t)dir * NKB * a.NB * 1024, (unsigned)NKB * kb_bytes);
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const unsigned vo = bval[i] ? (unsigned)(slice * BPC + wave + NW * i) * 2048u + (unsigned)lane * 16u : kOOBwq;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
                asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:1024"
                             : "=a"(whi[i][kb]), "=a"(wlo[i][kb]) : "v"(vo), "s"(wrs), "s"((unsigned)kb * kb_bytes) : "memory");
        }
        // every reader of these registers is an MFMA of the time loop, whose B operand comes out of LDS behind a workgroup barrier
        asm volatile("s_waitcnt vmcnt(0)" : "+a"(whi[0][0]), "+a"(wlo[BPW - 1][NKB - 1]) : : "memory");
    }

    // ---- xproj: rows tile-time-major, this cluster's NG 16-line groups are NG consecutive tiles of T*16 rows.  HBM -> LDS
    // directly; lanes without a valid row (finished lines, absent blocks, a group past N) read row 0 of the cluster's first tile
    const int ntiles = min((a.N - n0 + 15) / 16, NG);
    const i32x4 xrs = wq_srd(a.xp + (size_t)n0 * a.T * a.xstride + (size_t)dir * a.G, (unsigned)((size_t)ntiles * 16 * a.T * a.xstride * 4));
    const unsigned x_lane = (unsigned)line * (unsigned)a.xstride * 4u + (unsigned)us * 16u;   // this lane's 16 bytes inside a row of 16 lines
    const unsigned x_row = 16u * (unsigned)a.xstride * 4u;                                    // bytes per (tile, step)
    unsigned xso[BPW];                       // byte offset of the wave's blocks inside a line's row (wave-uniform: soffset)
#pragma unroll
    for (int i = 0; i < BPW; ++i) xso[i] = __builtin_amdgcn_readfirstlane(bval[i] ? (unsigned)(slice * BPC + wave + NW * i) * 64u : 0u);
    int x_len[NG];                           // the length of this lane's line in both groups (0: the group lies past N)
#pragma unroll
    for (int g = 0; g < NG; ++g) x_len[g] = g < ntiles ? lens_s[16 * g + line] : 0;
    auto load_x = [&](int g, int s) {
        const int t = rev ? (x_len[g] - 1 - s) : s;
        const unsigned vo = x_lane + (s < x_len[g] ? (unsigned)(g * a.T + t) * x_row : 0u);
        const unsigned l0 = __builtin_amdgcn_readfirstlane(xs_off + (unsigned)(((g & 1) * NW + wave) * BPW) * 1024u);
        unsigned keep;                       // M0 = the LDS base of a copy; saved, stepped per block, restored
        if constexpr (BPW == 1)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(xrs), "s"(l0), "s"(xso[0]) : "memory");
        else if constexpr (BPW == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(xrs), "s"(l0), "s"(xso[0]), "s"(xso[BPW > 1 ? 1 : 0]) : "memory", "scc");
        else if constexpr (BPW == 3)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %6 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(xrs), "s"(l0), "s"(xso[0]), "s"(xso[BPW > 1 ? 1 : 0]), "s"(xso[BPW > 2 ? 2 : 0]) : "memory", "scc");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %6 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %7 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(xrs), "s"(l0), "s"(xso[0]), "s"(xso[BPW > 1 ? 1 : 0]), "s"(xso[BPW > 2 ? 2 : 0]), "s"(xso[BPW > 3 ? 3 : 0])
                         : "memory", "scc");
    };

    // ---- exchange: granules [group][parity][slice][unit pair BPC*2][16 lines][2 units] of this cluster, 8 bytes each
    const unsigned slice_gran = (unsigned)BPC * 64u;              // granules one slice publishes per (group, step)
    const unsigned gp_bytes = 4u * slice_gran * 8u;               // bytes per (group, parity)
    const i32x4 grs = wq_srd(reinterpret_cast<const unsigned char*>(a.gran) + (size_t)cluster * (2 * NG) * gp_bytes, (unsigned)(2 * NG) * gp_bytes);
    const unsigned tagbase = (a.epoch & 0xFFFFu) << 16;
    auto gran_of = [&](int sl, int ul, int ln) -> unsigned {      // byte offset of the granule of (slice, local unit, line)
        return ((unsigned)sl * slice_gran + (((unsigned)(ul >> 1) * 16u + (unsigned)ln) * 2u + (unsigned)(ul & 1))) * 8u;
    };
    // what this lane publishes for block i: unit_local = (wave + NW i)*4 + us, its own line
    unsigned pub_vo[BPW], own_lds[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int ul = (wave + NW * i) * 4 + us;
        const int unit = slice * BPC * 4 + ul;
        pub_vo[i] = bval[i] ? gran_of(slice, ul, line) : kOOBwq;
        own_lds[i] = (bval[i] && unit < NKB * 32) ? lds_of(line, unit) : dump_off;
    }
    // HEARTBEAT (lstm_ws.hip): a slice without a gate-column block publishes one payload-free pair per (group, step) -- units 0 and
    // 1, line 0 of its own region -- and every peer gathers it (to nowhere): the flow control of the two parity buffers needs it
    const bool empty_slice = slice * BPC >= a.NB;
    if (empty_slice && wave == 0 && line == 0 && us < 2) pub_vo[0] = gran_of(slice, us, 0);
    // what this lane gathers: pair q = tid + NT k over [peer 3][BPC*2 unit pairs][16 lines]
    const unsigned slice_pairs = slice_gran >> 1;
    unsigned g_vo[NGP], g_lds[NGP], g_need[NGP];   // granule byte offset (kOOBwq: none), LDS byte offset of the pair's hi dword (dump strip: nowhere), all ones if the pair's tags count
#pragma unroll
    for (int k = 0; k < NGP; ++k) {
        const unsigned q = (unsigned)tid + (unsigned)NT * k;
        const unsigned p = q / slice_pairs, rem = q - p * slice_pairs;
        const int sl = (slice + 1 + (int)p) & 3;
        const int ul = 2 * (int)(rem >> 4), ln = (int)(rem & 15);
        const int unit = sl * BPC * 4 + ul;
        const bool beat = p < 3 && sl * BPC >= a.NB && rem == 0;     // the heartbeat pair of a slice without blocks
        const bool ok = (p < 3 && (sl * BPC + (ul >> 2)) < a.NB) || beat;      // blocks beyond NB are never published
        g_vo[k] = ok ? gran_of(sl, ul, ln) : kOOBwq;
        g_lds[k] = (ok && !beat && unit < NKB * 32) ? lds_of(ln, unit) : dump_off;
        g_need[k] = ok ? 0xFFFFFFFFu : 0u;
    }
    u32x4 gd[NGP];
    bool dead = false;
    // `after`: a value the first load pretends to read, so that the request cannot be scheduled before it exists
    // `on` false (the very first slot: there is no earlier step): every lane reads beyond the descriptor, zeros come back and
    // land in a buffer that holds zeros -- no branch in the slot
    auto gather_issue = [&](int g, int par, bool on = true, float after = 0.f) {
        const unsigned so = on ? (unsigned)(g * 2 + par) * gp_bytes : kOOBwq;
        asm volatile("; after %0" : : "v"(after));   // (an EMPTY asm string drops its operand)
#pragma unroll
        for (int k = 0; k < NGP; ++k) wq_load_b128_sc1(gd[k], g_vo[k], grs, so);
    };
    auto gather_drop = [&](unsigned char* hb) {       // granule payloads -> LDS rows: one byte permute per plane, one two-dword LDS write per pair
#pragma unroll
        for (int k = 0; k < NGP; ++k) {
            unsigned char* dst = hb + g_lds[k];
            const unsigned v0 = gd[k][0], v1 = gd[k][2];          // (hi | lo << 16) of units u, u + 1
            *reinterpret_cast<unsigned*>(dst) = __builtin_amdgcn_perm(v1, v0, 0x05040100u);
            *reinterpret_cast<unsigned*>(dst + planeP) = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
        }
    };
    auto gather_ok = [&](unsigned want) -> bool {    // vector ALU only: a compare per pair drags the scalar unit into a dependent chain
        unsigned bad = 0u;
#pragma unroll
        for (int k = 0; k < NGP; ++k) bad |= ((gd[k][1] ^ want) | (gd[k][3] ^ want)) & g_need[k];
        return bad == 0u;
    };
    static_assert(NGP >= 2 && NGP <= 6, "the waits below name every gather register");
    // waits until at most N younger vector-memory operations are outstanding; names every landing register, so that no reader of
    // them is scheduled above the wait
#define WQ_GATHER_WAIT(N)                                                                                                                  \
    do {                                                                                                                                   \
        if constexpr (NGP == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(gd[0]), "+v"(gd[1]) : "n"(N) : "memory");                        \
        else if constexpr (NGP == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(gd[0]), "+v"(gd[1]), "+v"(gd[2]) : "n"(N) : "memory");      \
        else if constexpr (NGP == 5) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(gd[0]), "+v"(gd[1]), "+v"(gd[2]), "+v"(gd[3]), "+v"(gd[4]) : "n"(N) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%6)" : "+v"(gd[0]), "+v"(gd[1]), "+v"(gd[2]), "+v"(gd[3]), "+v"(gd[4]), "+v"(gd[NGP - 1]) : "n"(N) : "memory"); \
    } while (0)
    // waits for the gather loads (the BPW publish stores were issued after them), then, branch-free: true if some granule of
    // h(g, step) did not carry its tag yet; the payloads go to LDS either way
    auto gather_try = [&](int step, unsigned char* hb) -> bool {
        WQ_GATHER_WAIT(BPW);
        const bool ok = gather_ok(tagbase | ((unsigned)(step + 1) & 0xFFFFu));
        gather_drop(hb);
        return !ok;
    };
    // slow path: polls until every granule of h(g, step) carries its tag, then drops it into LDS buffer hb
    auto gather_poll = [&](int g, int par, int step, unsigned char* hb) {
        const unsigned want = tagbase | ((unsigned)(step + 1) & 0xFFFFu);
        unsigned spins = 0;
        while (!dead) {
            gather_issue(g, par);
            WQ_GATHER_WAIT(0);
            if (__all(gather_ok(want))) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 21)) {          // ~ a second: give up, flag the plan, never wait again
                dead = true;
                if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        gather_drop(hb);
    };

    // ---- output pass, one 16-byte piece per lane (lstm_ws.hip).  Tile-time-major rows: this slice takes every fourth (plane, piece)
    // combination for all 16 lines of the group; line-major rows: lines [4*slice, 4*slice+4) of a group
    const int per_line = a.H >> 3;
    const size_t rows_total = a.otiled ? (size_t)((a.N + 15) / 16 * 16) * a.T : (size_t)a.N * a.T;
    const i32x4 ors = wq_srd(a.out, (unsigned)((size_t)a.out_plane * 4));
    unsigned sp_lds, sp_g00, sp_tmul;
    int sp_ln;
    if (a.otiled) {
        const int cmb = slice + 4 * (tid >> 4);                 // (plane, piece) combination, plane-major: 2*per_line <= 64 of them
        const int pl = cmb / per_line, q = cmb - pl * per_line;
        sp_ln = cmb < 2 * per_line ? (tid & 15) : -1;
        sp_lds = sp_ln >= 0 ? (unsigned)(pl * planeP + (q & 3) * OS + sp_ln * RSO + (q >> 2) * 16) : 0u;
        sp_g00 = (unsigned)((((size_t)(dir * per_line + q)) * rows_total + (size_t)(n0 >> 4) * a.T * 16 + (size_t)max(sp_ln, 0)) * 16 + (size_t)pl * a.out_plane * 2);
        sp_tmul = 16u * 16u;
    } else {
        const int e = tid;
        const int pl = e / (4 * per_line), r = e - pl * 4 * per_line;
        const int li = r / per_line, q = r - li * per_line;
        sp_ln = e < 8 * per_line ? slice * 4 + li : -1;
        sp_lds = sp_ln >= 0 ? (unsigned)(pl * planeP + (q & 3) * OS + sp_ln * RSO + (q >> 2) * 16) : 0u;
        sp_g00 = (unsigned)((((size_t)(dir * per_line + q)) * rows_total + (size_t)(n0 + max(sp_ln, 0)) * a.T) * 16 + (size_t)pl * a.out_plane * 2);
        sp_tmul = 16u;
    }
    const unsigned sp_gmul = 16u * (unsigned)a.T * 16u;
    int sp_len[NG];                          // the length of the line this lane writes, in both groups (0: none)
#pragma unroll
    for (int g = 0; g < NG; ++g) sp_len[g] = sp_ln >= 0 ? lens_s[16 * g + sp_ln] : 0;
    auto store_read = [&](int g, int step, const unsigned char* hb, unsigned& vo) -> u32x4 {
        const int len = sp_len[g];
        const bool on = step >= 0 && step < len;
        const int t = rev ? (len - 1 - step) : step;
        vo = on ? sp_g00 + (unsigned)g * sp_gmul + (unsigned)t * sp_tmul : kOOBwq;
        return *reinterpret_cast<const u32x4*>(hb + sp_lds);
    };

    float cst[NG][BPW];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < BPW; ++i) cst[g][i] = 0.f;
#ifdef KRK_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#endif

    // one slot = one time step of one group.  `pend`: this wave's gather of h(g, s-1) met a stale tag (rare);
    // `nxt`: start the gather of h(ng, nxt_step) (parity nxt_par), which the NEXT slot needs
    bool pend[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) pend[g] = false;
    auto slot = [&](int g, int s, bool nxt, int ng, int nxt_par, int nxt_step) {
        const int par = s & 1;
        unsigned char* hb = hs + (g * 2 + par) * hbuf;            // h(g, s-1): own rows written by our gates, the rest gathered
        unsigned char* hn = hs + (g * 2 + (par ^ 1)) * hbuf;      // h(g, s)
        if (__any(pend[g]) && !KRK_DBGBIT(a, 1)) { WQ_COUNT(7); gather_poll(g, par, s - 1, hb); }
        pend[g] = false;
        WQ_STAMP(5);                                  // [5] gather poll (slow path) + whatever sits between the slots
        if (!KRK_DBGBIT(a, 32)) __syncthreads();
        WQ_STAMP(0);                                  // [0] barrier wait
        // xproj of this slot was requested two slots ago; since then this wave issued at least the BPW publish stores and the
        // BPW xproj requests of the previous slot: everything older has landed
        wq_wait<2 * BPW>();
        f32x4 xv[BPW];
#pragma unroll
        for (int i = 0; i < BPW; ++i) xv[i] = *reinterpret_cast<const f32x4*>(smem8 + xs_off + (((g & 1) * NW + wave) * BPW + i) * 1024 + lane * 16);
        unsigned sp_vo = kOOBwq;
        u32x4 sp_v = u32x4{0u, 0u, 0u, 0u};
        if (!KRK_DBGBIT(a, 16)) sp_v = store_read(g, s - 1, hb, sp_vo);
        bf16x8 hh[NKB], hl[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const unsigned char* hp = hb + us * OS + line * RSO + kb * 16;
            hh[kb] = *reinterpret_cast<const bf16x8*>(hp);
            hl[kb] = *reinterpret_cast<const bf16x8*>(hp + planeP);
        }
        if (!KRK_DBGBIT(a, 16)) wq_store_b128(sp_v, sp_vo, ors);
        const unsigned want = tagbase | ((unsigned)(s + 1) & 0xFFFFu);
        const unsigned pso = (unsigned)(g * 2 + (par ^ 1)) * gp_bytes;
        // ONE wave per SIMD: the matrix pipe and the vector ALU overlap only if the instruction stream interleaves them (a lump of
        // nine VALU instructions behind an MFMA triplet idles the pipe for 24 of 72 cycles: tools/ubench/mfma_agpr_probe).  The
        // slot is one straight-line region; the sched_group_barriers behind it ask for MFMA, two VALU, MFMA, two VALU, ...: the
        // cell update of block i (a dependent chain of ~45 instructions) then trickles into the MFMA stream of block i + 1.
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            f32x4 acc0 = xv[i], acc1 = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                wq_pin(whi[i][kb], wlo[i][kb]);
                if (KRK_DBGBIT(a, 4)) continue;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq_bf(whi[i][kb]), hh[kb], acc0, 0, 0, 0);
                KRK_CROSS(acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq_bf(whi[i][kb]), hl[kb], acc1, 0, 0, 0);
                          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq_bf(wlo[i][kb]), hh[kb], acc2, 0, 0, 0);)
            }
            asm("" : "+v"(acc0), "+v"(acc1), "+v"(acc2));               // accumulators in VGPRs: the VALU cannot read the other half
            // the peers finished publishing the other group about when this slot began and a granule needs ~0.5 us to become
            // visible: ask for them a block's worth of MFMAs into the slot, look at them at its end
            if (i == 0 && !KRK_DBGBIT(a, 1)) gather_issue(ng, nxt_par, nxt, acc0[0] + acc1[0] + acc2[0]);
            if (KRK_DBGBIT(a, 2)) continue;
            const f32x4 z = acc0 + (acc1 + acc2);
            const float h = krk_lstm_cell(z, cst[g][i]);          // 7 transcendentals per unit (common.h)
            const __bf16 hb16 = (__bf16)h;
            const __bf16 lb16 = (__bf16)(h - (float)hb16);
            const unsigned short hbits = __builtin_bit_cast(unsigned short, hb16), lbits = __builtin_bit_cast(unsigned short, lb16);
            unsigned char* dst = hn + own_lds[i];
            *reinterpret_cast<unsigned short*>(dst) = hbits;
            *reinterpret_cast<unsigned short*>(dst + planeP) = lbits;
            u32x2 gran;
            gran[0] = (unsigned)hbits | ((unsigned)lbits << 16);
            gran[1] = want;
            wq_store_b64(gran, pub_vo[i], grs, pso);
        }
#pragma unroll
        for (int n = 0; n < 3 * NKB * BPW; ++n) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);    // two VALU / transcendental instructions
        }
        WQ_STAMP(3);                                  // [3] the last block's cell update (exposed)
        if (!KRK_DBGBIT(a, 1)) pend[ng] = gather_try(nxt_step, hs + (ng * 2 + nxt_par) * hbuf) && nxt;   // optimistic finish of the next slot's gather
        WQ_STAMP(4);                                  // [4] wait for the gather loads + tag check + LDS rows
    };

    // ---- prologue.  xproj of a slot is requested two slots ahead
    load_x(0, 0);
    load_x(1, 0);
    wq_wait<0>();                                     // the counted wait inside a slot assumes a full previous slot behind it
    for (int s = 0; s < Lmax; ++s) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // the next slot: (g+1, s) needs h(g+1, s-1), parity s&1 -- or (0, s+1) (also the epilogue) needs h(0, s), parity (s+1)&1
            if (g + 1 < NG) slot(g, s, s > 0, g + 1, s & 1, s - 1);
            else slot(g, s, true, 0, (s + 1) & 1, s);
            if (!KRK_DBGBIT(a, 8)) load_x((g + 2) % NG, s + (g + 2) / NG);
        }
    }
    if (Lmax > 0) {
        const int par = Lmax & 1;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            unsigned char* hb = hs + (g * 2 + par) * hbuf;
            if (g == 0) {
                if (__any(pend[0]) && !KRK_DBGBIT(a, 1)) gather_poll(0, par, Lmax - 1, hb);
            } else if (!KRK_DBGBIT(a, 1)) {
                gather_poll(g, par, Lmax - 1, hb);
            }
            wq_wait<0>();
            __syncthreads();
            unsigned vo;
            const u32x4 v = store_read(g, Lmax - 1, hb, vo);
            wq_store_b128(v, vo, ors);
        }
    }
#ifdef KRK_STAMP
    if (a.stamps && cluster == 0 && slice == 0 && lane == 0) {
        st_acc[6] = (unsigned long long)Lmax * NG;
        for (int i = 0; i < 8; ++i) a.stamps[wave * 8 + i] = st_acc[i];
    }
#endif
}

template <int NKB, int BPW, int NW>
int launch_wq(const LstmWsArgs& a, hipStream_t s) {
    const int nclusters = a.nclusters;
    const size_t lds = (((size_t)2 * 2 * 2 * (4 * 16 * 16 * (NKB | 1) + 256 * NW) + 32 * sizeof(int) + 16 + 256) & ~(size_t)255) + (size_t)2 * NW * BPW * 1024;
    auto kfn = lstm_wq_kernel<NKB, BPW, NW>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3((unsigned)nclusters * 4 + 24), dim3(64 * NW), lds, s, a);     // 8 x 3 surplus workgroups: see the cluster claim
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

#ifndef KRK_BF16_ONE
// ctrl block of a launch (zeroed by the host before every launch): [0..7] arrivals per XCD, [8] work items claimed, [9] posted,
// [16 + x * mbox + i] mailbox of the i-th local group of XCD x; mbox = groups one XCD can see if EVERY block lands on it
int krk_lstm_wq_mbox(int nclusters) { return nclusters + 7; }
size_t krk_lstm_wq_ctrl_bytes(int nclusters) { return (size_t)(16 + 8 * krk_lstm_wq_mbox(nclusters)) * 4; }
#endif

// a.wp = the streaming kernel's fragments (upload_lstm_x3: [dir][kb][block][plane][lane][8]); everything else as krk_launch_lstm_ws
int KRK_FN(krk_launch_lstm_wq)(const LstmWsArgs& a, int waves, hipStream_t s) {
    if (!krk_lstm_ws_supported(a.H, a.Hp)) return -4;
    if ((size_t)a.out_plane * 4 >= 0x80000000ull) return -4;                 // 32-bit buffer offsets
    if (a.T >= 0xFFFF) return -4;                                             // 16-bit step tags
    if (a.nclusters != (a.N + 31) / 32 * a.ndir || a.mbox != krk_lstm_wq_mbox(a.nclusters)) return -4;
    // NB = Hp/4 in (8(NKB-1), 8 NKB]; BPC = ceil(NB/4) in {2 NKB - 1, 2 NKB}; BPW = ceil(BPC / waves)
    if (waves == 8) {
        const int bpw = (a.BPC + 7) / 8;
#define KRK_WQ(NKB_, BPW_) if (a.NKB == NKB_ && bpw == BPW_) return launch_wq<NKB_, BPW_, 8>(a, s)
        KRK_WQ(1, 1); KRK_WQ(2, 1); KRK_WQ(3, 1); KRK_WQ(4, 1); KRK_WQ(5, 2); KRK_WQ(6, 2); KRK_WQ(7, 2);
#undef KRK_WQ
        return -4;
    }
    const int bpw = (a.BPC + 3) / 4;
#define KRK_WQ(NKB_, BPW_) if (a.NKB == NKB_ && bpw == BPW_) return launch_wq<NKB_, BPW_, 4>(a, s)
    KRK_WQ(1, 1); KRK_WQ(2, 1); KRK_WQ(3, 2); KRK_WQ(4, 2); KRK_WQ(5, 3); KRK_WQ(6, 3); KRK_WQ(7, 4);
#undef KRK_WQ
    return -4;
}
