// Weight-stationary recurrent kernel of the (bi)directional LSTM: split-bf16 operands on the bf16 matrix cores, W_hh held
// in REGISTERS by a cluster of four workgroups (one per CU) that exchange h_t through L2 every step.
// Reference semantics: nn.LSTM inside TransposedSummarizingRNN.forward (kraken/lib/vgsl/layers.py:513-547): packed by
// length, gates i,f,g,o, h/c start at zero, outputs past a line's length stay zero.
//
// Why: the streaming kernels (lstm_x3.hip) pull the whole W_hh (hi+lo = the bytes of fp32: 0.7 MB for H = 200) through one
// CU's L1 every time step -- 5.2 us at the 135 GB/s a CU gets from L2 -- and the HBM-latency xproj loads share that
// in-order L1 queue, so the two serialise (measured: 9.1 us per step, matrix pipe 20 % busy).  Here
//   * a cluster = 4 workgroups x 8 waves (two per SIMD, <= 256 registers: no AGPR copies in front of the MFMAs).  Workgroup
//     `slice` owns a quarter of the gate-column blocks (all four gates of its units, so the cell update stays per lane), a
//     wave owns up to BPW of them and keeps their fragments for all K blocks in registers for the whole launch: no weight
//     traffic after the prologue;
//   * a cluster advances TWO independent groups of 16 lines in alternation ("slots"): while the h_t of one group travels
//     (publish -> L2 -> the three peers), the matrix pipe works on the other group;
//   * exchange = data-tagged 8-byte granules {tag = launch epoch | step+1, (hi, lo) bf16 of one (unit, line)} written with
//     one sc1 (write-through, agent-scope) store and polled with sc1 loads: the data is the flag, no fence, no separate
//     counter, placement independent (MI355X guide, Guideline 16 form R2).  Two parity buffers per group suffice: a slice
//     can only publish step s+2 after it consumed every peer's step s+1, which they publish after consuming step s;
//   * cluster membership is claimed at run time (ticket = atomicAdd): any four workgroups that have STARTED form a
//     cluster, so a partially resident grid cannot deadlock whatever the dispatch order; every spin is bounded and a
//     timeout raises the plan's error word (mapped host memory) instead of hanging the device;
//   * the gather loads of the NEXT slot's group are issued one block of MFMAs into a slot (the peers published that group about
//     when the slot began; a granule needs ~0.5 us to become visible) and checked optimistically at its end (branch-free: the
//     granules go to LDS whatever their tag; a lane that saw a stale tag only raises a flag), so the common case costs no wait
//     at all; a flagged gather is polled at the start of the next slot.  A lane gathers PAIRS of adjacent units of one line:
//     two 8-byte loads -> one dword LDS write per plane;
//   * inside a slot LDS returns in order: the small reads (xproj landing, output piece) first, then the 2*NKB fragments in the
//     order the MFMAs consume them, so the matrix pipe starts on the first pair while the other waves' reads are queued;
//   * every vector-memory instruction of the time loop is inline assembly with hand-counted `s_waitcnt vmcnt(N)`: vmcnt
//     retires in order and an sc1 (write-through) publish store takes ~1 us to be acknowledged, so a wait that is one count
//     too strict stalls a slot behind the stores it just issued -- which is what the compiler's bookkeeping (conservative
//     across the loop's branches) produced: 2.07 us per slot, 0.56 of it MFMA.  xproj lands in LDS (global_load_lds) two
//     slots ahead, so no register is live across the loop edge while its load is in flight;
//   * h_t leaves as K-blocked split planes for the next projection (gemm_x3.hip): after the gather every slice holds the
//     whole h_t in LDS.  Rows tile-time-major (a.otiled, the rows between recurrent layers): a slice writes every fourth
//     (plane, piece) combination for all 16 lines of the group -- whole 256-byte runs; line-major rows (other consumers): a
//     quarter of the lines, 16 bytes per lane, each in a different 128-byte line.  Masked by the buffer bounds check.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned kOOBws = 0x80000000u;   // voffset beyond every descriptor used here (all < 2 GiB): load = 0, store dropped

__device__ __forceinline__ bf16x8 ws_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// ---- vector memory by hand (see the header comment): the compiler neither sees nor counts these
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 ws_srd(const void* p, unsigned bytes) {       // raw buffer descriptor in SGPRs
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    i32x4 r;
    r[0] = (int)__builtin_amdgcn_readfirstlane((unsigned)u);
    r[1] = (int)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xFFFFu);
    r[2] = (int)__builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void vm_load_b64_sc1(u32x2& d, unsigned vo, const i32x4& srd, unsigned so) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen sc1" : "=&v"(d) : "v"(vo), "s"(srd), "s"(so) : "memory");
}
__device__ __forceinline__ void vm_store_b64_sc1(const u32x2& d, unsigned vo, const i32x4& srd, unsigned so) {
    asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen sc1" : : "v"(d), "v"(vo), "s"(srd), "s"(so) : "memory");
}
__device__ __forceinline__ void vm_store_b128(const u32x4& d, unsigned vo, const i32x4& srd) {
    // s_nop: wait states of the ">64-bit store data, then VALU write of those registers" hazard (invisible to the compiler here)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(d), "v"(vo), "s"(srd) : "memory");
}
__device__ __forceinline__ void vm_load_lds_b128(const float* gptr, unsigned lds_off) {   // 64 lanes x 16 B -> LDS [lds_off, +1 KB)
    unsigned keep;                                                                        // M0 (LDS base of the copy) is saved and restored
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_off) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// NKB K blocks of 32, BPW gate-column blocks per wave, NG groups of 16 lines per cluster (slots per time step; even)
template <int NKB, int BPW, int NG>
__global__ void __launch_bounds__(512) lstm_ws_kernel(const LstmWsArgs a) {
    // a lane gathers NGP PAIRS of granules per (group, step): two adjacent units of one line -> one dword per plane in LDS
    // (3 peers x BPC*32 pairs / 512 lanes, BPC <= 8*BPW)
    constexpr int NGP = (3 * BPW + 1) / 2;
    constexpr int NGI = 2 * NGP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    // h in LDS, per (group, parity): [plane hi|lo][K octet 4][line 16][K block: 16 bytes each, padded to an odd count].
    // A fragment read (lane = line + 16*octet) then touches 16 distinct lines of at most two octet planes per 16-lane
    // group: conflict free.  (The [line][K] rows of the streaming kernel put two octets of different lines on the same
    // banks: PMC showed 44 % of this kernel's LDS cycles as bank conflicts, all 8 waves reading right after the barrier.)
    constexpr int RSO = 16 * (NKB | 1);     // bytes per (octet, line) row
    constexpr int OS = 16 * RSO;            // one octet plane
    constexpr int plane = 4 * OS;           // hi / lo plane
    constexpr int hbuf = 2 * plane;         // one (hi, lo) buffer of one group
    auto lds_of = [&](int ln, int unit) -> unsigned {        // byte offset of (line, unit) inside a buffer's hi plane
        return (unsigned)(((unit & 31) >> 3) * OS + ln * RSO + (unit >> 5) * 16 + (unit & 7) * 2);
    };
    unsigned char* hs = smem8;              // [group NG][parity 2][hbuf]
    int* lens_s = reinterpret_cast<int*>(smem8 + 2 * NG * hbuf);        // [16 * NG]
    unsigned* misc = reinterpret_cast<unsigned*>(lens_s + 16 * NG);     // [0] ticket
    // masked LDS writes (absent blocks, padding granules) land in a dump area, one dword per lane: 64 lanes hammering ONE
    // dump word were the kernel's bank conflicts (PMC: 44 % of its LDS cycles, unchanged by the h layout)
    const unsigned dump_base = (unsigned)(2 * NG * hbuf + 16 * NG * 4 + 16);
    const unsigned xs_off = dump_base + 256;                             // xproj landing ring [slot parity 2][wave 8][BPW][64 lanes x 16 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) misc[0] = atomicAdd(a.ctrl, 1u) - a.ticket_base;
    for (int e = tid; e < NG * hbuf / 2; e += 512) reinterpret_cast<unsigned int*>(hs)[e] = 0u;   // 2*NG*hbuf bytes
    __syncthreads();
    const unsigned ticket = __builtin_amdgcn_readfirstlane(misc[0]);
    const int cluster = (int)(ticket >> 2), slice = (int)(ticket & 3);
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
    const unsigned dump_off = dump_base + (unsigned)lane * 4u;
    const int BPC = a.BPC;

    // ---- weights: resident for the whole launch.  [dir][slice][wave 8][i][kb][plane][lane][8]
    u32x4 whi[BPW][NKB], wlo[BPW][NKB];
    {
        const __bf16* wb = a.wp + ((((size_t)dir * 4 + slice) * 8 + wave) * BPW * NKB) * 1024 + lane * 8;
#pragma unroll
        for (int i = 0; i < BPW; ++i)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                whi[i][kb] = *reinterpret_cast<const u32x4*>(wb + (size_t)(i * NKB + kb) * 1024);
                wlo[i][kb] = *reinterpret_cast<const u32x4*>(wb + (size_t)(i * NKB + kb) * 1024 + 512);
            }
    }
    // which of this wave's blocks exist (wave-uniform): local block wave + 8i < BPC, global block < NB
    bool bval[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) bval[i] = (wave + 8 * i < BPC) && (slice * BPC + wave + 8 * i < a.NB);

    // ---- xproj: rows are tile-time-major, this cluster's NG 16-line groups are NG consecutive tiles of T*16 rows.
    // HBM -> LDS directly (no bounds check on this path: lanes without a valid row -- finished lines, absent blocks, a group
    // past N -- read row 0 of the cluster's first tile; whatever they compute stays in their own MFMA column / dump word).
    const int ntiles = min((a.N - n0 + 15) / 16, NG);
    const float* xbase = a.xp + (size_t)n0 * a.T * a.xstride + (size_t)dir * a.G + (size_t)line * a.xstride + us * 4;
    unsigned xcol[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) xcol[i] = bval[i] ? (unsigned)(slice * BPC + wave + 8 * i) * 16u : 0u;
    // `ring`: the landing buffer = parity of the slot that will consume it (slot index = s*NG + g, NG even: g & 1)
    auto load_x = [&](int g, int s) {
        const int len = lens_s[16 * g + line];
        const int t = rev ? (len - 1 - s) : s;
        const size_t row = (s < len && g < ntiles) ? ((size_t)g * a.T + t) * 16 : 0;
        const float* xr = xbase + row * a.xstride;
#pragma unroll
        for (int i = 0; i < BPW; ++i) vm_load_lds_b128(xr + xcol[i], xs_off + (unsigned)((((g & 1) * 8 + wave) * BPW + i) * 1024));
    };

    // ---- exchange: granules [group][parity][slice][BPC*4 units][16 lines] of this cluster
    const unsigned slice_gran = (unsigned)BPC * 64u;              // granules one slice publishes per (group, step)
    const unsigned gp_bytes = 4u * slice_gran * 8u;               // bytes per (group, parity)
    const i32x4 grs = ws_srd(reinterpret_cast<const unsigned char*>(a.gran) + (size_t)cluster * (2 * NG) * gp_bytes, (unsigned)(2 * NG) * gp_bytes);
    const unsigned tagbase = (a.epoch & 0xFFFFu) << 16;
    // what this lane publishes for block i: unit_local = (wave + 8i)*4 + us, its own line
    unsigned pub_vo[BPW], own_lds[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int ul = (wave + 8 * i) * 4 + us;
        const int unit = slice * BPC * 4 + ul;
        const bool ok = bval[i];
        pub_vo[i] = ok ? ((unsigned)slice * slice_gran + (unsigned)ul * 16u + (unsigned)line) * 8u : kOOBws;
        own_lds[i] = (ok && unit < NKB * 32) ? lds_of(line, unit) : dump_off;
    }
    // HEARTBEAT (round 4: the cause of the exchange timeouts on narrow layers).  The two parity buffers are safe only because a
    // producer cannot publish step s+2 before it has gathered step s+1 from every peer -- which that peer publishes after ALL its
    // waves consumed step s.  A slice WITHOUT a gate-column block (slice * BPC >= NB: NB in {1, 2, 3, 5, 6, 9}, i.e. hidden sizes
    // below 40) published nothing, so nobody ever waited for it: its peers ran ahead, overwrote h(g, s) with h(g, s+2) before it had
    // read it, and it spun on a tag that was gone (5..50 timeouts in 100 forwards of an H = 8 net; never at H >= 40, where every
    // slice owns a block).  An empty slice now publishes one pair of payload-free granules per (group, step) -- units 0 and 1,
    // line 0 of its own, otherwise unused region -- and every peer gathers that pair (to nowhere): the same flow control for all.
    const bool empty_slice = slice * BPC >= a.NB;
    if (empty_slice && wave == 0 && line == 0 && us < 2) pub_vo[0] = ((unsigned)slice * slice_gran + (unsigned)us * 16u) * 8u;
    // what this lane gathers: pair q = tid + 512 k over [peer 3][BPC*2 unit pairs][16 lines]; packed into one register: bit 31 =
    // wanted, bits 16..30 = LDS offset / 4 of the pair's dword (0x7FFF = nowhere: padding units), bits 0..15 = index of the
    // first granule (the second one is 16 granules = one unit further)
    const unsigned slice_pairs = slice_gran >> 1;
    unsigned g_item[NGP];
#pragma unroll
    for (int k = 0; k < NGP; ++k) {
        const unsigned q = (unsigned)tid + 512u * k;
        const unsigned p = q / slice_pairs, rem = q - p * slice_pairs;
        const int sl = (slice + 1 + (int)p) & 3;
        const int ul = 2 * (int)(rem >> 4), ln = (int)(rem & 15);
        const int unit = sl * BPC * 4 + ul;
        const bool beat = p < 3 && sl * BPC >= a.NB && rem == 0;     // the heartbeat pair of a slice without blocks (see above)
        const bool ok = (p < 3 && (sl * BPC + (ul >> 2)) < a.NB) || beat;      // blocks beyond NB are never published
        const unsigned lo = (ok && !beat && unit < NKB * 32) ? (lds_of(ln, unit) >> 2) : 0x7FFFu;
        g_item[k] = (ok ? 0x80000000u : 0u) | (lo << 16) | (((unsigned)sl * slice_gran + (unsigned)ul * 16u + (unsigned)ln) & 0xFFFFu);
    }
    auto g_vo = [&](int k) -> unsigned { return (g_item[k >> 1] >> 31) ? ((g_item[k >> 1] & 0xFFFFu) + 16u * (k & 1)) * 8u : kOOBws; };
    u32x2 gd[NGI];
    bool dead = false;
    // `after`: a value the first load pretends to read, so that the request cannot be scheduled before it exists
    auto gather_issue = [&](int g, int par, float after = 0.f) {
        const unsigned so = (unsigned)(g * 2 + par) * gp_bytes;
        asm volatile("; after %0" : : "v"(after));   // (an EMPTY asm string drops its operand: the request then floats to the slot start)
#pragma unroll
        for (int k = 0; k < NGI; ++k) vm_load_b64_sc1(gd[k], g_vo(k), grs, so);
    };
    auto gather_drop = [&](unsigned char* hb) {       // granule payloads -> LDS rows (masked pairs go to the lane's dump word)
#pragma unroll
        for (int k = 0; k < NGP; ++k) {
            const unsigned lo = (g_item[k] >> 16) & 0x7FFFu;
            const bool nowhere = lo == 0x7FFFu;
            unsigned char* dst = nowhere ? smem8 + dump_off : hb + 4 * lo;
            const unsigned v0 = gd[2 * k][0], v1 = gd[2 * k + 1][0];          // (hi | lo << 16) of units u, u + 1
            *reinterpret_cast<unsigned*>(dst) = (v0 & 0xFFFFu) | (v1 << 16);
            *reinterpret_cast<unsigned*>(dst + (nowhere ? 0 : plane)) = (v0 >> 16) | (v1 & 0xFFFF0000u);
        }
    };
    // waits for the gather loads (BPW publish stores were issued after them), then, branch-free: true if some granule of
    // h(g, step) did not carry its tag yet; the payloads go to LDS either way
    static_assert((NGI == 4 && BPW == 1) || (NGI == 6 && BPW == 2), "the waits below count BPW publish stores behind NGI loads");
    auto gather_try = [&](int step, unsigned char* hb) -> bool {
        const unsigned want = tagbase | ((unsigned)(step + 1) & 0xFFFFu);
        if constexpr (NGI == 4) asm volatile("s_waitcnt vmcnt(1)" : "+v"(gd[0]), "+v"(gd[1]), "+v"(gd[2]), "+v"(gd[3]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(2)" : "+v"(gd[0]), "+v"(gd[1]), "+v"(gd[2]), "+v"(gd[3]), "+v"(gd[4]), "+v"(gd[5]) : : "memory");
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NGI; ++k) ok = ok && (!(g_item[k >> 1] >> 31) || gd[k][1] == want);
        gather_drop(hb);
        return !ok;
    };
    // slow path: polls until every granule of h(g, step) carries its tag, then drops it into LDS buffer hb
    auto gather_poll = [&](int g, int par, int step, unsigned char* hb) {
        const unsigned want = tagbase | ((unsigned)(step + 1) & 0xFFFFu);
        unsigned spins = 0;
        while (!dead) {
            gather_issue(g, par);
            if constexpr (NGI == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(gd[0]), "+v"(gd[1]), "+v"(gd[2]), "+v"(gd[3]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(gd[0]), "+v"(gd[1]), "+v"(gd[2]), "+v"(gd[3]), "+v"(gd[4]), "+v"(gd[5]) : : "memory");
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NGI; ++k) ok = ok && (!(g_item[k >> 1] >> 31) || gd[k][1] == want);
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 21)) {          // ~ a second: give up, flag the plan, never wait again
                dead = true;
                if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        gather_drop(hb);
    };

    // ---- output pass, one 16-byte piece per lane.  Line-major rows: this slice writes lines [4*slice, 4*slice+4) of a group (200
    // different 128-byte lines per slot: a piece's rows of one step lie T rows apart).  Tile-time-major rows (a.otiled: the
    // consumer is gemm_x3, which keeps the order): the 16 lines of a group at one step are 16 consecutive rows, so this slice
    // takes every fourth (plane, piece) combination for ALL 16 lines -- whole 256-byte runs, 26 lines per slot.
    const int per_line = a.H >> 3;
    const size_t rows_total = a.otiled ? (size_t)((a.N + 15) / 16 * 16) * a.T : (size_t)a.N * a.T;
    const i32x4 ors = ws_srd(a.out, (unsigned)((size_t)a.out_plane * 4));
    unsigned sp_lds, sp_g00, sp_tmul;
    int sp_ln;
    if (a.otiled) {
        const int cmb = slice + 4 * (tid >> 4);                 // (plane, piece) combination, plane-major
        const int pl = cmb / per_line, q = cmb - pl * per_line;
        sp_ln = cmb < 2 * per_line ? (tid & 15) : -1;
        sp_lds = sp_ln >= 0 ? (unsigned)(pl * plane + (q & 3) * OS + sp_ln * RSO + (q >> 2) * 16) : 0u;
        sp_g00 = (unsigned)((((size_t)(dir * per_line + q)) * rows_total + (size_t)(n0 >> 4) * a.T * 16 + (size_t)max(sp_ln, 0)) * 16 + (size_t)pl * a.out_plane * 2);
        sp_tmul = 16u * 16u;                                      // bytes per time step: 16 rows
    } else {
        const int e = tid;
        const int pl = e / (4 * per_line), r = e - pl * 4 * per_line;
        const int li = r / per_line, q = r - li * per_line;
        sp_ln = e < 8 * per_line ? slice * 4 + li : -1;
        sp_lds = sp_ln >= 0 ? (unsigned)(pl * plane + (q & 3) * OS + sp_ln * RSO + (q >> 2) * 16) : 0u;   // piece q = units 8q..8q+7
        sp_g00 = (unsigned)((((size_t)(dir * per_line + q)) * rows_total + (size_t)(n0 + max(sp_ln, 0)) * a.T) * 16 + (size_t)pl * a.out_plane * 2);
        sp_tmul = 16u;                                            // consecutive steps of a line are consecutive rows
    }
    // group g of the cluster: line-major 16 lines = 16*T rows further, tile-time-major the next tile = 16*T rows further too
    const unsigned sp_gmul = 16u * (unsigned)a.T * 16u;
    // two halves: the LDS read is issued FIRST in a slot (LDS returns in order: a store that waited for a read issued after the
    // 14 fragment reads would hold the wave -- and its first MFMA -- until the whole fragment set had arrived)
    auto store_read = [&](int g, int step, const unsigned char* hb, unsigned& vo) -> u32x4 {
        const int len = sp_ln >= 0 ? lens_s[16 * g + sp_ln] : 0;
        const bool on = step >= 0 && step < len;
        const int t = rev ? (len - 1 - step) : step;
        vo = on ? sp_g00 + (unsigned)g * sp_gmul + (unsigned)t * sp_tmul : kOOBws;
        return *reinterpret_cast<const u32x4*>(hb + sp_lds);
    };
    auto store_pass = [&](int g, int step, const unsigned char* hb) {
        unsigned vo;
        const u32x4 v = store_read(g, step, hb, vo);
        vm_store_b128(v, vo, ors);
    };

    float cst[NG][BPW];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < BPW; ++i) cst[g][i] = 0.f;

    // one slot = one time step of one group.  `pend`: this wave's gather of h(g, s-1) met a stale tag (rare);
    // `nxt`: start the gather of h(ng, nxt_step) (parity nxt_par), which the NEXT slot needs
    bool pend[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) pend[g] = false;
    auto slot = [&](int g, int s, bool nxt, int ng, int nxt_par, int nxt_step) {
        const int par = s & 1;
        unsigned char* hb = hs + (g * 2 + par) * hbuf;            // h(g, s-1): own rows written by our gates, the rest gathered
        unsigned char* hn = hs + (g * 2 + (par ^ 1)) * hbuf;      // h(g, s)
        if (__any(pend[g]) && !KRK_DBGBIT(a, 1)) gather_poll(g, par, s - 1, hb);
        pend[g] = false;
        if (!KRK_DBGBIT(a, 32)) __syncthreads();
        // xproj of this slot was requested two slots ago; since then this wave issued at least the BPW publish stores and the
        // BPW xproj requests of the previous slot (its exchange loads were consumed there): everything older has landed
        vm_wait<2 * BPW>();
        // LDS returns in order: the small reads first (xproj landing, the output piece of the previous step), then the 2*NKB
        // fragments in the order the MFMAs consume them, so that the matrix pipe starts on the first pair while the other
        // waves' reads are still queued (8 waves x 14 KB = 875 LDS cycles per slot)
        f32x4 xv[BPW];
#pragma unroll
        for (int i = 0; i < BPW; ++i) xv[i] = *reinterpret_cast<const f32x4*>(smem8 + xs_off + (((g & 1) * 8 + wave) * BPW + i) * 1024 + lane * 16);
        unsigned sp_vo = kOOBws;
        u32x4 sp_v = u32x4{0u, 0u, 0u, 0u};
        if (!KRK_DBGBIT(a, 16)) sp_v = store_read(g, s - 1, hb, sp_vo);
        // NKB == 8 (hidden sizes 225 ... 256, round 6): 128 registers of resident weights leave no room for all sixteen h fragments
        // of a slot (the compiler spilled eight registers, and its scratch traffic would sit between the hand-counted vmcnt waits):
        // the fragments are read K block by K block, each feeding BOTH gate-column blocks of the wave, and the two cell updates
        // follow the last MFMA.  Per accumulator the K blocks still arrive in ascending order.
        constexpr bool KMAJOR = NKB >= 8;
        bf16x8 hh[KMAJOR ? 1 : NKB], hl[KMAJOR ? 1 : NKB];
        if constexpr (!KMAJOR) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const unsigned char* hp = hb + us * OS + line * RSO + kb * 16;
            hh[kb] = *reinterpret_cast<const bf16x8*>(hp);
            hl[kb] = *reinterpret_cast<const bf16x8*>(hp + plane);
        }
        }
        if (!KRK_DBGBIT(a, 16)) vm_store_b128(sp_v, sp_vo, ors);
        // NG > 2: the next slot's h was published NG-1 slots ago -- ask for it now, it lands under this slot's MFMAs
        if ((NG > 2 || KRK_DBGBIT(a, 64)) && nxt && !KRK_DBGBIT(a, 1)) gather_issue(ng, nxt_par);
        const unsigned want = tagbase | ((unsigned)(s + 1) & 0xFFFFu);
        const unsigned pso = (unsigned)(g * 2 + (par ^ 1)) * gp_bytes;
        if constexpr (KMAJOR) {
            f32x4 acc0[BPW], acc1[BPW], acc2[BPW];
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                acc0[i] = xv[i];
                acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const unsigned char* hp = hb + us * OS + line * RSO + kb * 16;
                const bf16x8 fh = *reinterpret_cast<const bf16x8*>(hp);
                const bf16x8 fl = *reinterpret_cast<const bf16x8*>(hp + plane);
#pragma unroll
                for (int i = 0; i < BPW; ++i) {
                    acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_bf(whi[i][kb]), fh, acc0[i], 0, 0, 0);
                    KRK_CROSS(acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_bf(whi[i][kb]), fl, acc1[i], 0, 0, 0);
                              acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_bf(wlo[i][kb]), fh, acc2[i], 0, 0, 0);)
                }
                if (NG == 2 && kb == NKB / 2 - 1 && nxt && !KRK_DBGBIT(a, 1)) gather_issue(ng, nxt_par, acc0[0][0] + acc1[0][0] + acc2[0][0]);
            }
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const f32x4 z = acc0[i] + (acc1[i] + acc2[i]);
                const float h = krk_lstm_cell(z, cst[g][i]);
                const __bf16 hb16 = (__bf16)h;
                const __bf16 lb16 = (__bf16)(h - (float)hb16);
                const unsigned short hbits = __builtin_bit_cast(unsigned short, hb16), lbits = __builtin_bit_cast(unsigned short, lb16);
                unsigned char* dst = (own_lds[i] == dump_off) ? smem8 + dump_off : hn + own_lds[i];
                *reinterpret_cast<unsigned short*>(dst) = hbits;
                *reinterpret_cast<unsigned short*>(dst + ((own_lds[i] == dump_off) ? 2 : plane)) = lbits;
                u32x2 gran;
                gran[0] = (unsigned)hbits | ((unsigned)lbits << 16);
                gran[1] = want;
                vm_store_b64_sc1(gran, pub_vo[i], grs, pso);
            }
        } else
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            f32x4 acc0 = xv[i], acc1 = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (KRK_DBGBIT(a, 4)) break;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_bf(whi[i][kb]), hh[kb], acc0, 0, 0, 0);
                KRK_CROSS(acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_bf(whi[i][kb]), hl[kb], acc1, 0, 0, 0);
                          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_bf(wlo[i][kb]), hh[kb], acc2, 0, 0, 0);)
            }
            // NG == 2: the peers finished publishing the other group about when this slot began and a granule needs ~0.5 us to
            // become visible: ask for them a block's worth of MFMAs into the slot, look at them at its end
            if (NG == 2 && !KRK_DBGBIT(a, 64) && i == (KRK_DBGBIT(a, 128) ? BPW - 1 : 0) && nxt && !KRK_DBGBIT(a, 1)) gather_issue(ng, nxt_par, acc0[0] + acc1[0] + acc2[0]);
            if (KRK_DBGBIT(a, 2)) continue;
            const f32x4 z = acc0 + (acc1 + acc2);
            const float h = krk_lstm_cell(z, cst[g][i]);          // 7 transcendentals per unit instead of 10 (common.h)
            const __bf16 hb16 = (__bf16)h;
            const __bf16 lb16 = (__bf16)(h - (float)hb16);
            const unsigned short hbits = __builtin_bit_cast(unsigned short, hb16), lbits = __builtin_bit_cast(unsigned short, lb16);
            unsigned char* dst = (own_lds[i] == dump_off) ? smem8 + dump_off : hn + own_lds[i];
            *reinterpret_cast<unsigned short*>(dst) = hbits;
            *reinterpret_cast<unsigned short*>(dst + ((own_lds[i] == dump_off) ? 2 : plane)) = lbits;
            u32x2 gran;
            gran[0] = (unsigned)hbits | ((unsigned)lbits << 16);
            gran[1] = want;
            vm_store_b64_sc1(gran, pub_vo[i], grs, pso);
        }
        if (nxt && !KRK_DBGBIT(a, 1)) pend[ng] = gather_try(nxt_step, hs + (ng * 2 + nxt_par) * hbuf);   // optimistic finish of the next slot's gather
    };

    // ---- prologue.  xproj of a slot is requested two slots ahead
    load_x(0, 0);
    load_x(1, 0);
    vm_wait<0>();                                     // the counted wait inside a slot assumes a full previous slot behind it
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
            vm_wait<0>();
            __syncthreads();
            store_pass(g, Lmax - 1, hb);
        }
    }
}

template <int NKB, int BPW, int NG>
int launch_ws(const LstmWsArgs& a, hipStream_t s) {
    const int nclusters = (a.N + 16 * NG - 1) / (16 * NG) * a.ndir;
    const size_t lds = (size_t)2 * NG * 2 * 4 * 16 * 16 * (NKB | 1) + 16 * NG * sizeof(int) + 16 + 256 + (size_t)2 * 8 * BPW * 1024;
    auto kfn = lstm_ws_kernel<NKB, BPW, NG>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3((unsigned)nclusters * 4), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

#ifndef KRK_BF16_ONE
bool krk_lstm_ws_supported(int H, int Hp) {
    const int NKB = (Hp + 31) / 32, NB = Hp / 4;
    return (H % 8) == 0 && NKB >= 1 && NKB <= 8 && NB <= 64 && (NB + 3) / 4 <= 16;
}

int krk_lstm_ws_clusters(int N, int ndir, int groups) { return (N + 16 * groups - 1) / (16 * groups) * ndir; }

size_t krk_lstm_ws_gran_bytes(int N, int ndir, int BPC, int groups) {
    return (size_t)krk_lstm_ws_clusters(N, ndir, groups) * (2 * groups) /* (group, parity) */ * 4 /* slices */ * (size_t)BPC * 64 * 8;
}

#endif

// groups = 16-line groups a cluster advances in alternation: 2 (32 lines per 4 CUs: lowest latency, the exchange round trip
// is about one slot) or 4 (64 lines per 4 CUs: half the CUs, the h of a group is three slots old when it is read)
int KRK_FN(krk_launch_lstm_ws)(const LstmWsArgs& a, int groups, hipStream_t s) {
    if (!krk_lstm_ws_supported(a.H, a.Hp)) return -4;
    if ((size_t)a.out_plane * 4 >= 0x80000000ull) return -4;                 // 32-bit buffer offsets
    if (a.T >= 0xFFFF) return -4;                                             // 16-bit step tags
    if (3 * a.BPC * 64 > 0xFFFF) return -4;                                   // 16-bit granule index
    const int bpw = (a.BPC + 7) / 8;
#define KRK_WS(NKB_, BPW_) if (a.NKB == NKB_ && bpw == BPW_) return groups == 4 ? launch_ws<NKB_, BPW_, 4>(a, s) : launch_ws<NKB_, BPW_, 2>(a, s)
    // NB = Hp/4 in (8(NKB-1), 8 NKB]; BPC = ceil(NB/4) in {2 NKB - 1, 2 NKB}; BPW = ceil(BPC/8)
    KRK_WS(1, 1); KRK_WS(2, 1); KRK_WS(3, 1); KRK_WS(4, 1); KRK_WS(5, 2); KRK_WS(6, 2); KRK_WS(7, 2); KRK_WS(8, 2);
#undef KRK_WS
    return -4;
}
