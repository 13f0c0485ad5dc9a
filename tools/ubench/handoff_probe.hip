// Dev probe: which (store flavour, load flavour) pairs hand an 8-byte {payload, tag} granule from one CU to another on the SAME
// XCD (blocks 0 and 8) and on different XCDs (blocks 0 and 1), and how long does a hop take?  Ping-pong of one lane per block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int S> __device__ __forceinline__ void st(unsigned long long* p, u32x2 v) {
    if constexpr (S == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if constexpr (S == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if constexpr (S == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (S == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (S == 4) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
}
template <int L> __device__ __forceinline__ u32x2 ld(const unsigned long long* p) {
    u32x2 v;
    if constexpr (L == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (L == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (L == 2) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (L == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (L == 4) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// blocks a and b ping-pong `rounds` times through slots[0] (a -> b) and slots[16] (b -> a); out[0] = cycles, out[1] = rounds completed
template <int S, int L>
__global__ void pingpong(unsigned long long* slots, int a, int b, int rounds, unsigned epoch, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    unsigned long long* mine = slots + (me == 0 ? 0 : 16);
    const unsigned long long* theirs = slots + (me == 0 ? 16 : 0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    int done = 0;
    for (int r = 1; r <= rounds; ++r) {
        const unsigned tag = (epoch << 16) | (unsigned)r;
        if (me == 0) st<S>(mine, u32x2{(unsigned)r * 3u, tag});
        unsigned spins = 0;
        bool ok = false;
        while (spins++ < 200000u) { u32x2 v = ld<L>(theirs); if (v[1] == tag) { ok = true; break; } }
        if (!ok) break;
        if (me == 1) st<S>(mine, u32x2{(unsigned)r * 5u, tag});
        done = r;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[me * 2] = t1 - t0;
    out[me * 2 + 1] = (unsigned long long)done;
}
template <int S, int L> void run(unsigned long long* slots, unsigned long long* out, int b, unsigned& epoch, const char* sn, const char* ln) {
    unsigned long long h[4];
    hipMemset(out, 0, 32);
    ++epoch;
    hipLaunchKernelGGL((pingpong<S, L>), dim3(16), dim3(64), 0, 0, slots, 0, b, 2000, epoch, out);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("  store %-8s load %-8s %s: rounds %4llu/2000, %7.1f cycles per hop (100 MHz counter -> x10 ns)\n", sn, ln, b == 8 ? "same XCD " : "cross XCD",
           h[1], h[1] ? (double)h[0] / (2.0 * h[1]) : 0.0);
}
#define ROW(S, SN) \
    run<S, 0>(slots, out, b, epoch, SN, "sc1"); run<S, 1>(slots, out, b, epoch, SN, "sc0 sc1"); run<S, 2>(slots, out, b, epoch, SN, "nt"); \
    run<S, 3>(slots, out, b, epoch, SN, "sc0"); run<S, 4>(slots, out, b, epoch, SN, "plain");
int main() {
    unsigned long long *slots, *out;
    hipMalloc(&slots, 4096); hipMalloc(&out, 64);
    hipMemset(slots, 0, 4096);
    unsigned epoch = 0;
    for (int b : {8, 1}) {
        ROW(0, "plain") ROW(1, "sc0") ROW(2, "sc1") ROW(3, "sc0 sc1") ROW(4, "nt")
    }
    return 0;
}
