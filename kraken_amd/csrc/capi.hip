// C ABI + plan executor of the MI355X recognition path (see include/kraken_amd.h).
//
// krk_plan_create plays the role of TorchVGSLModel._parse (reference
// kraken/lib/vgsl/model.py:202-243): it turns the layer list into a fused kernel
// schedule and repacks the state-dict tensors into MFMA fragment order.
// krk_forward plays the role of MultiParamSequential.forward (layers.py:44-53).
#include "common.h"
#include "../../include/kraken_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <memory>
#include <vector>

// phase cycle counters of the instrumented kernel (common.h KRK_PHASES): conv_x3p; returns the count
int krk_phase_stats_x3p(unsigned long long* out, int reset);
#ifdef KRK_ABLATE
extern "C" int krk_debug_phase_stats(int which, unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return which == 0 ? krk_phase_stats_x3p(out, reset) : -1;
}
#endif

namespace {

thread_local std::string g_err;
#ifdef KRK_ABLATE
#endif

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(KRK_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

enum StepKind { S_CONV = 0, S_MAXPOOL, S_GN, S_TOSEQ, S_LSTM, S_LINEAR, S_IMG2ROWS, S_ROWS2IMG, S_UNSPLIT, S_ALIAS, S_CONCAT, S_ADD, S_SOFTMAXC, S_UPZERO, S_PERMUTE };
const char* kStepNames[] = {"conv", "maxpool", "groupnorm", "to_seq", "lstm", "linear", "img2rows", "rows2img", "unsplit", "alias", "concat", "add", "softmax", "zero_insert", "reshape"};



static int env_int(const char* name, int dflt = 0) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// roctx ranges around a call and its launch groups ("roctx ranges around submit / forward / decode").  The marker
// library is NOT a link dependency: it is looked up once -- already in the process (rocprofv3 --marker-trace preloads
// librocprofiler-sdk-roctx.so) or, with KRK_ROCTX=1, opened by name (then libroctx64.so for the older tools) -- and without it
// every call below is one predictable branch.
#include <dlfcn.h>
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
        const bool want = env_int("KRK_ROCTX") != 0;
        void* h = nullptr;
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!h && want)
            for (const char* n : names)
                if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h || getenv("KRK_NO_ROCTX")) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
    bool on() const { return push != nullptr; }
};
static const Roctx& roctx() {
    static Roctx r;
    return r;
}
// one range, closed when it goes out of scope
struct RoctxRange {
    bool open = false;
    explicit RoctxRange(const std::string& name) {
        if (roctx().on()) { roctx().push(name.c_str()); open = true; }
    }
    ~RoctxRange() { if (open) roctx().pop(); }
};

// Python-style floor division (the reference does float division + floor).
inline int floordiv(int a, int b) {
    int q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) {
            if (hipDeviceSynchronize() != hipSuccess) return -1;
            (void)hipFree(p);
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 8;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            return -1;
        }
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Geometry + packed weights of one implicit-GEMM (conv / projection) launch.
struct ConvGeom {
    int Cin = 0, H = 1, Cout = 0;
    int kh = 1, kw = 1, sh = 1, sw = 1, dh = 1, dw = 1, ph = 0, pw = 0;
    int act = 0;
    bool in_seq = false, out_seq = false, pool = false;
    int Ho = 1, Hy = 1;
    int SR = 8, TH = 1, TW = 256, IH = 1, IW = 256, RS = 256, PS = 257;
    int cchunk = 1, nchunks = 1, Kc = 1, KSG = 1, KSG_last = 1, KSGpad = 4, KS4 = 16, vec4 = 0, CB = 1, CBpad = 1,
        otab_floats = 32;
    float* d_w = nullptr;
    float* d_b = nullptr;
    // split-bf16 ("bf16x3") execution of this GEMM (conv_x3.hip)
    bool x3 = false;          // run on the bf16 matrix cores with split operands
    bool split_out = false;   // write the output as split channels-last bf16 planes
    int xchunk = 16, xnchunks = 1, xKB = 1, xKB_last = 1, xPSTR = 48, xplane = 0;
    int xK = 0;               // gemm_x3.hip: K of the projection = Cin rounded up to 16 (round 6: the rows' last octet is a zeroed pad when Cin % 16 == 8)
    int xtps = 0;             // > 0: the pipelined kernel (conv_x3p.hip) covers this geometry: tile copies per weight-stage boundary
    bool x6 = false;          // three-plane ("bf16x6") convolution in front of a GroupNorm (conv_x6.hip): fp32 NCHW in and out
    void* d_wx6 = nullptr;    // its weights [chunk][tap][block][plane 3][lane][8]
    int x6CBpad = 1;
    void* d_wx3 = nullptr;
    void* d_wx5 = nullptr;    // conv_taps_x3.hip, five-group packing (kw <= 13)
    bool c1x3 = false;        // one-channel first convolution on the bf16 cores (conv1_x3.hip); weights in d_wx3
    bool taps = false;        // wide-kernel convolution with taps as K (conv_taps_x3.hip); reads NHCW planes
    bool out_nhcw = false;    // conv1_x3 writes [N][H][C][pitch] planes for a following taps convolution
};

// row pitch (elements) of the "NHCW" planes between conv1_x3 and conv_taps_x3: whole 16-byte pieces
int nhcw_pitch(int w) { return (w + 63) / 64 * 64; }   // whole 128-byte lines per (row, channel): a column tile of conv1_x3 stores exactly one line (with the 16-byte pitch of round 1 every 128-byte store straddled two lines: 615 MB written for 472)

int conv_out(int L, int k, int s, int d, int p) { return floordiv(L + 2 * p - d * (k - 1) - 1, s) + 1; }

// Chooses the tile shape / channel chunking for a conv whose input height is known.
void plan_conv_geom(ConvGeom& g) {
    g.CB = (g.Cout + 31) / 32;
    const int cbw = g.CB >= 4 ? 4 : (g.CB >= 2 ? 2 : 1);
    g.CBpad = (g.CB + cbw - 1) / cbw * cbw;
    if (g.in_seq) {
        g.SR = 8;
        g.TH = 1;
        g.TW = 256;
        g.IH = 1;
        g.IW = g.TW;
        g.RS = g.TW + 1;
        g.PS = g.TW + 1;
        g.Ho = g.Hy = 1;
    } else {
        g.Ho = conv_out(g.H, g.kh, g.sh, g.dh, g.ph);
        g.Hy = g.pool ? floordiv(g.Ho - 2, 2) + 1 : g.Ho;
        if (g.Ho >= 3) g.SR = 2;
        else if (g.Ho == 2 || g.pool) g.SR = 4;
        else g.SR = 8;
        // 5..8 output rows without a pool: one 8-row tile (fewer dead rows than two 4-row tiles; conv4 of BENCH-A 0.235 -> 0.215 ms)
        if (!g.pool && g.Ho > 4 && g.Ho <= 8) g.SR = 1;
        g.TH = 8 / g.SR;
        g.TW = 32 * g.SR;
        g.IH = (g.TH - 1) * g.sh + (g.kh - 1) * g.dh + 1;
        g.IW = (g.TW - 1) * g.sw + (g.kw - 1) * g.dw + 1;
        g.RS = g.IW;
        g.PS = g.IH * g.IW;
    }
    const int kk = g.kh * g.kw;
    // largest channel chunk whose LDS tile stays within 64 KiB (= NPT * 256 staged floats per workgroup)
    // 48 KB (round 6; was 64): three workgroups of the exact-f32 kernel share a CU where two did (BENCH-A f32 plan: the 32 -> 32, 3 x 13 layer 4.6 -> 3.4 ms
    // under load; KRK_F32_TILE_KB, read when the plan is built, is the probe: profiles/r06_occupancy_three_workgroups.txt)
    const int kTileFloats = std::min(64, std::max(8, env_int("KRK_F32_TILE_KB", 48))) * 256;
    int cmax;
    g.vec4 = 0;
    if (g.in_seq && g.Cin % 4 == 0) {
        g.vec4 = 1;
        cmax = std::min(64, g.Cin);                       // 16 lanes x 4 features per pixel row
    } else {
        cmax = std::max(1, std::min(kTileFloats / g.PS, g.Cin));
    }
    g.nchunks = (g.Cin + cmax - 1) / cmax;
    g.cchunk = g.vec4 ? cmax : (g.Cin + g.nchunks - 1) / g.nchunks;
    g.Kc = g.cchunk * kk;
    g.KSG = ((g.Kc + 1) / 2 + 3) / 4;
    const int kc_last = (g.Cin - (g.nchunks - 1) * g.cchunk) * kk;
    g.KSG_last = ((kc_last + 1) / 2 + 3) / 4;
    g.KSGpad = g.KSG + 3;                                 // slack read by the software pipeline
    g.KS4 = 4 * g.KSGpad;
    g.otab_floats = 2 * g.KS4;
}

// wpack[chunk][group][cb][lane][e] = W[cout = cb*32 + (lane&31)][k = 2*(4*group + e) + (lane>>5)], k -> (c, dy, dx)
// `rowmap` (optional) maps packed output column -> source row of `w` (or -1 for a zero column).
int upload_conv_weights(ConvGeom& g, const float* w, const float* bias, const std::vector<int>* rowmap,
                        const std::vector<float>* bias_override) {
    const int kk = g.kh * g.kw;
    if (!g.in_seq && (size_t)g.cchunk * g.PS > 64 * 256)
        return fail(KRK_E_UNSUPPORTED, "convolution window too large for the LDS tile");
    std::vector<float> pack((size_t)g.nchunks * g.KSGpad * g.CBpad * 256, 0.f);
    for (int ci = 0; ci < g.nchunks; ++ci)
        for (int gr = 0; gr < g.KSG; ++gr)
            for (int cb = 0; cb < g.CB; ++cb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int co = cb * 32 + (lane & 31);
                        const int k = 2 * (4 * gr + e) + (lane >> 5);
                        if (co >= g.Cout || k >= g.Kc) continue;
                        const int cl = k / kk, rem = k % kk;
                        const int c = ci * g.cchunk + cl;
                        if (c >= g.Cin) continue;
                        int src = co;
                        if (rowmap) {
                            src = (*rowmap)[co];
                            if (src < 0) continue;
                        }
                        pack[((((size_t)ci * g.KSGpad + gr) * g.CBpad + cb) * 64 + lane) * 4 + e] =
                            w[((size_t)src * g.Cin + c) * kk + rem];
                    }
    std::vector<float> b((size_t)g.CBpad * 32, 0.f);
    for (int co = 0; co < g.Cout; ++co) {
        if (bias_override) b[co] = (*bias_override)[co];
        else if (bias) b[co] = bias[rowmap ? std::max((*rowmap)[co], 0) : co];
    }
    HIPCHK(hipMalloc((void**)&g.d_w, pack.size() * sizeof(float)));
    HIPCHK(hipMemcpy(g.d_w, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&g.d_b, b.size() * sizeof(float)));
    HIPCHK(hipMemcpy(g.d_b, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    return KRK_OK;
}

// bf16 round-to-nearest-even (matches v_cvt_pk_bf16_f32 for finite values)
inline uint16_t f2bf(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);   // inf / nan
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// LDS tile geometry of the split-bf16 kernel: channels per chunk (multiple of 16) such that both planes of
// the (IH x IW) pixel tile stay within ~74 KiB (two workgroups per CU).
int plan_x3_geom(ConvGeom& g) {
    if (g.Cin % 16) return fail(KRK_E_UNSUPPORTED, "bf16x3: input channels/features must be a multiple of 16");
    const int npix = g.IH * g.IW;
    int c = g.Cin;
    // 2 workgroups per CU: 160 KB / 2 minus the 24 KB weight ring of conv_x3.hip
    const size_t budget = (size_t)env_int("KRK_X3_LDS_KB", 52) * 1024;
    while (c > 16 && (size_t)2 * npix * (c * 2 + 16) > budget) c -= 16;
    if ((size_t)2 * npix * (c * 2 + 16) > 130 * 1024) return fail(KRK_E_UNSUPPORTED, "bf16x3: convolution window too large");
    g.xnchunks = (g.Cin + c - 1) / c;
    g.xchunk = ((g.Cin + g.xnchunks - 1) / g.xnchunks + 15) / 16 * 16;
    g.xnchunks = (g.Cin + g.xchunk - 1) / g.xchunk;
    g.xKB = g.xchunk / 16;
    g.xKB_last = (g.Cin - (g.xnchunks - 1) * g.xchunk) / 16;
    g.xPSTR = g.xchunk * 2 + 16;
    g.xplane = npix * g.xPSTR;
    // conv_x3p.hip (asynchronous double-buffered staging) works on 16-channel chunks: where it covers the geometry the chunking
    // is its own -- conv_x3.hip runs the same plan (same weights) when the probe switch KRK_CONV_X3P=0 asks for it
    if (!getenv("KRK_NO_CONV_X3P")) {
        const int tps = krk_conv_x3p_tps(16, 1, 1, npix, g.IW, g.kh * g.kw, g.Cout, 0);
        if (tps > 0) {
            g.xtps = tps;
            g.xchunk = 16;
            g.xnchunks = g.Cin / 16;
            g.xKB = g.xKB_last = 1;
            g.xPSTR = 48;
            g.xplane = npix * g.xPSTR;
        }
    }
    return KRK_OK;
}

// wx3[chunk][tap][kb][cb][plane][lane][8] (bf16): lane l of block cb holds filter cb*32 + (l&31), channels
// chunk*xchunk + kb*16 + 8*(l>>5) + 0..7 of tap (dy,dx); plane 0 = hi, 1 = lo.  `w` is (rows, Cin, kh, kw) f32.
int upload_x3_weights(ConvGeom& g, const float* w, const std::vector<int>* rowmap) {
    const int kk = g.kh * g.kw;
    const int cb = krk_x3_cb(g.Cout);
    const int CBt = (g.Cout + 31) / 32;
    const int CBpad = (CBt + cb - 1) / cb * cb;
    if (CBpad != g.CBpad) return fail(KRK_E_INVALID, "bf16x3: filter block padding mismatch");
    // + 8 zero records of slack: the kernel's weight cursor prefetches up to 4 records past the end
    std::vector<uint16_t> pack(((size_t)g.xnchunks * kk * g.xKB + 8) * CBpad * 1024, 0);
    for (int ci = 0; ci < g.xnchunks; ++ci) {
        const int kbn = (ci + 1 == g.xnchunks) ? g.xKB_last : g.xKB;   // the last chunk is packed densely
        for (int t = 0; t < kk; ++t)
            for (int kb = 0; kb < kbn; ++kb)
                for (int b = 0; b < CBt; ++b)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = b * 32 + (lane & 31);
                            const int c = ci * g.xchunk + kb * 16 + 8 * (lane >> 5) + e;
                            if (co >= g.Cout || c >= g.Cin) continue;
                            int src = co;
                            if (rowmap) {
                                src = (*rowmap)[co];
                                if (src < 0) continue;
                            }
                            const float v = w[((size_t)src * g.Cin + c) * kk + t];
                            const uint16_t hi = f2bf(v);
                            const uint16_t lo = f2bf(v - bf2f(hi));
                            const size_t rec = (size_t)ci * kk * g.xKB + (size_t)t * kbn + kb;
                            const size_t base = (rec * CBpad + b) * 1024 + lane * 8 + e;
                            pack[base] = hi;
                            pack[base + 512] = lo;
                        }
    }
    HIPCHK(hipMalloc(&g.d_wx3, pack.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(g.d_wx3, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return KRK_OK;
}

// conv_x6.hip: 16-channel chunks, tile = 3 planes x npix x 48 bytes, ring 3 x (cb x 3 KB): two workgroups per CU
int plan_x6_geom(ConvGeom& g) {
    if (g.Cin % 16) return -1;
    const int npix = g.IH * g.IW;
    const int cb = krk_x6_cb(g.Cout);
    if ((size_t)3 * npix * 48 + (size_t)9 * cb * 1024 > 80 * 1024) return -1;
    g.xchunk = 16;
    g.xnchunks = g.Cin / 16;
    g.xKB = g.xKB_last = 1;
    g.xPSTR = 48;
    g.xplane = npix * 48;
    const int CBt = (g.Cout + 31) / 32;
    g.x6CBpad = (CBt + cb - 1) / cb * cb;
    return 0;
}

// wx6[chunk][tap][cb][plane 3][lane][8] (bf16): lane l of block cb holds filter cb*32 + (l&31), channels chunk*16 + 8*(l>>5) + 0..7 of
// tap (dy,dx); planes h = bf16(w), m = bf16(w - h), l = bf16(w - h - m).  `w` is (Cout, Cin, kh, kw) f32.
int upload_x6_weights(ConvGeom& g, const float* w) {
    const int kk = g.kh * g.kw;
    const int CBt = (g.Cout + 31) / 32, CBpad = g.x6CBpad;
    std::vector<uint16_t> pack(((size_t)g.xnchunks * kk + 4) * CBpad * 1536, 0);     // + slack records (the ring looks two stages ahead)
    for (int ci = 0; ci < g.xnchunks; ++ci)
        for (int t = 0; t < kk; ++t)
            for (int b = 0; b < CBt; ++b)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int co = b * 32 + (lane & 31);
                        const int c = ci * 16 + 8 * (lane >> 5) + e;
                        if (co >= g.Cout || c >= g.Cin) continue;
                        const float v = w[((size_t)co * g.Cin + c) * kk + t];
                        const uint16_t h = f2bf(v);
                        const float r1 = v - bf2f(h);
                        const uint16_t m = f2bf(r1);
                        const uint16_t l = f2bf(r1 - bf2f(m));
                        const size_t base = ((((size_t)ci * kk + t) * CBpad + b) * 3) * 512 + lane * 8 + e;
                        pack[base] = h;
                        pack[base + 512] = m;
                        pack[base + 1024] = l;
                    }
    HIPCHK(hipMalloc(&g.d_wx6, pack.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(g.d_wx6, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return KRK_OK;
}

// conv1_x3.hip weight order: [channel][kernel row dy][plane][lane][8]; lane = filter + 32*half holds taps 8*half..+7.  `w` is
// (Cout, Cin, kh, kw), Cin = 1 or 3.
// Round 6: more than 32 filters (a first layer of 64: kraken specs that open with Cr3,3,64) = one such pack per 32 filters, one launch each.
int upload_conv1_x3_weights(ConvGeom& g, const float* w) {
    const int halves = (g.Cout + 31) / 32;
    const size_t per_half = (size_t)g.Cin * g.kh * 2 * 64 * 8;
    std::vector<uint16_t> pack(per_half * halves, 0);
    for (int hf = 0; hf < halves; ++hf)
    for (int ch = 0; ch < g.Cin; ++ch)
        for (int dy = 0; dy < g.kh; ++dy)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int f = 32 * hf + (lane & 31), dx = 8 * (lane >> 5) + e;
                    if (f >= g.Cout || dx >= g.kw) continue;
                    const float v = w[(((size_t)f * g.Cin + ch) * g.kh + dy) * g.kw + dx];
                    const uint16_t hi = f2bf(v);
                    const size_t row = (size_t)(ch * g.kh + dy) * 2;
                    pack[hf * per_half + ((row + 0) * 64 + lane) * 8 + e] = hi;
                    pack[hf * per_half + ((row + 1) * 64 + lane) * 8 + e] = f2bf(v - bf2f(hi));
                }
    HIPCHK(hipMalloc(&g.d_wx3, pack.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(g.d_wx3, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return KRK_OK;
}

// conv_taps_x3.hip weight order: [channel][kernel row][plane][lane][8]; lane = filter + 32*half holds taps 8*half..+7
int upload_conv_taps_weights(ConvGeom& g, const float* w) {
    std::vector<uint16_t> pack((size_t)g.Cin * g.kh * 2 * 64 * 8, 0);
    for (int ch = 0; ch < g.Cin; ++ch)
        for (int dy = 0; dy < g.kh; ++dy)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int f = lane & 31, dx = 8 * (lane >> 5) + e;
                    if (f >= g.Cout || dx >= g.kw) continue;
                    const float v = w[(((size_t)f * g.Cin + ch) * g.kh + dy) * g.kw + dx];
                    const uint16_t hi = f2bf(v);
                    const size_t base = ((size_t)(ch * g.kh + dy) * 2) * 64 * 8 + (size_t)lane * 8 + e;
                    pack[base] = hi;
                    pack[base + 512] = f2bf(v - bf2f(hi));
                }
    HIPCHK(hipMalloc(&g.d_wx3, pack.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(g.d_wx3, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    // Five-group packing (conv_taps_x3.hip header): [channel pair][kernel row][fragment 4][lane][8]; lane = filter + 32*half,
    // half = channel of the pair; fragment 0 = w_hi taps 0..7, 1 = w_lo taps 0..7, 2 = w_hi 8..12 | w_hi 8..10,
    // 3 = w_hi 11..12 | w_lo 8..12 | 0.  KRK_NO_TAPS5 keeps the six-group kernel (A/B probing).
    if (g.kw > 13 || g.kh != 3 || g.Cin % 2 || getenv("KRK_NO_TAPS5")) return KRK_OK;
    std::vector<uint16_t> p5((size_t)(g.Cin / 2) * g.kh * 4 * 64 * 8, 0);
    for (int pr = 0; pr < g.Cin / 2; ++pr)
        for (int dy = 0; dy < g.kh; ++dy)
            for (int lane = 0; lane < 64; ++lane) {
                const int f = lane & 31, ch = 2 * pr + (lane >> 5);
                if (f >= g.Cout) continue;
                auto split = [&](int dx, uint16_t& hi, uint16_t& lo) {
                    hi = lo = 0;
                    if (dx >= g.kw) return;
                    const float v = w[(((size_t)f * g.Cin + ch) * g.kh + dy) * g.kw + dx];
                    hi = f2bf(v);
                    lo = f2bf(v - bf2f(hi));
                };
                uint16_t* q = &p5[(((size_t)(pr * g.kh + dy) * 4) * 64 + lane) * 8];
                const size_t FR = 64 * 8;
                for (int e = 0; e < 8; ++e) {
                    uint16_t hi, lo;
                    split(e, hi, lo);
                    q[0 * FR + e] = hi;
                    q[1 * FR + e] = lo;
                    split(e < 5 ? 8 + e : 8 + (e - 5), hi, lo);          // Gd: hi x hi taps 8..12, then hi x lo taps 8..10
                    q[2 * FR + e] = hi;
                    if (e < 2) { split(11 + e, hi, lo); q[3 * FR + e] = hi; }        // Ge: hi x lo taps 11..12 ...
                    else if (e < 7) { split(8 + (e - 2), hi, lo); q[3 * FR + e] = lo; }   // ... lo x hi taps 8..12, slot 7 idle
                }
            }
    HIPCHK(hipMalloc(&g.d_wx5, p5.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(g.d_wx5, p5.data(), p5.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return KRK_OK;
}

// gemm_x3.hip weight order: [column group of tn][K/16][plane][k-half][column][8] (bf16); `w` is (rows, K) f32.
int pack_gemm_x3_weights(const ConvGeom& g, const float* w, const std::vector<int>* rowmap, int tn, void** dst) {
    const int ncg = (g.Cout + tn - 1) / tn, nkb = g.Cin / 16;
    const size_t rec = (size_t)32 * tn;            // elements per (column group, K step): 2 planes x 2 k-halves x tn x 8
    std::vector<uint16_t> pack((size_t)ncg * nkb * rec, 0);
    for (int cg = 0; cg < ncg; ++cg)
        for (int kb = 0; kb < nkb; ++kb)
            for (int col = 0; col < tn; ++col) {
                const int co = cg * tn + col;
                if (co >= g.Cout) continue;
                int src = co;
                if (rowmap) {
                    src = (*rowmap)[co];
                    if (src < 0) continue;
                }
                for (int h = 0; h < 2; ++h)
                    for (int e = 0; e < 8; ++e) {
                        const float v = w[(size_t)src * g.Cin + kb * 16 + h * 8 + e];
                        const uint16_t hi = f2bf(v);
                        const size_t base = ((size_t)cg * nkb + kb) * rec + ((size_t)h * tn + col) * 8 + e;
                        pack[base] = hi;
                        pack[base + rec / 2] = f2bf(v - bf2f(hi));
                    }
            }
    HIPCHK(hipMalloc(dst, pack.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(*dst, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return KRK_OK;
}

// Round 6: K % 16 == 8 (2 x 100 hidden units: kraken's classic recognisers) -- the K-blocked rows get one more octet, zeroed per call,
// and the weights a zero column block; before, such a network lost its whole split-bf16 plan.
// `colmap` (+ `kphys` physical features): input feature c of the torch weights lies at column colmap[c] of the rows the producer
// writes (a recurrent layer whose directions are padded to Hp units each, Step::opad); the other columns are zero.
int upload_gemm_x3_weights(ConvGeom& g, const float* w, const std::vector<int>* rowmap, const std::vector<int>* colmap = nullptr, int kphys = 0) {
    // (any K since round 6: a row's last octet may be partly real -- a convolution stack's C x H that is not a multiple of 8 --
    // the rest of it and, for K % 16 in 1 .. 8, one more octet are zeroed per call; such a network had lost its whole split-bf16 plan)
    g.xK = ((colmap ? kphys : g.Cin) + 15) / 16 * 16;
    if (g.xK == g.Cin && !colmap) return pack_gemm_x3_weights(g, w, rowmap, 128, &g.d_wx3);
    int nrows = g.Cout;
    if (rowmap)
        for (int v : *rowmap) nrows = std::max(nrows, v + 1);
    std::vector<float> wp((size_t)nrows * g.xK, 0.f);
    for (int r = 0; r < nrows; ++r) {
        if (!colmap) { std::memcpy(&wp[(size_t)r * g.xK], w + (size_t)r * g.Cin, (size_t)g.Cin * sizeof(float)); continue; }
        for (int c = 0; c < g.Cin; ++c) wp[(size_t)r * g.xK + (*colmap)[c]] = w[(size_t)r * g.Cin + c];
    }
    ConvGeom gp = g;
    gp.Cin = g.xK;
    const int rc = pack_gemm_x3_weights(gp, wp.data(), rowmap, 128, &g.d_wx3);
    return rc;
}

struct Step {
    StepKind kind = S_CONV;
    ConvGeom cg;              // CONV / LINEAR / the LSTM input projection
    // shape of the step's input: NCHW (C,H) or sequence (C = features, H = 1)
    int C = 0, H = 1;
    // MAXPOOL
    int kh = 1, kw = 1, sh = 1, sw = 1, Ho = 1;
    bool pooled = false;      // S_GN: the MaxPool that follows is part of the step (kh .. Ho describe it)
    // GROUPNORM
    int groups = 1;
    float* d_gamma = nullptr;
    float* d_beta = nullptr;
    // c1gn.hip: the one-channel first convolution in front of this GroupNorm is recomputed inside its two passes (the conv
    // step is skipped: `skip`); taps [C][9] and bias [C] of that convolution
    bool skip = false, c1gn = false;
    float* d_c1w = nullptr;
    float* d_c1b = nullptr;
    int c1_act = 0;
    // LSTM
    int hidden = 0, Hp = 0, ndir = 1, dirmode = 0;
    float* d_wrec32 = nullptr;  // recurrent weights, 32x32x2 fragment order
    float* d_wrec16 = nullptr;  // recurrent weights, 16x16x4 fragment order
    void* d_wrecx3 = nullptr;   // recurrent weights, split bf16, 16x16x32 fragment order
    // weight-stationary cluster kernel (lstm_ws.hip): per-wave resident fragments, exchange granules, ticket counter
    void* d_wrecws = nullptr;
    DevBuf ws_gran;
    unsigned* ws_ctrl = nullptr;
    unsigned ws_tickets = 0, ws_epoch = 0;
    int ws_bpc = 0;
    float* d_wrecsm = nullptr;  // recurrent weights for lstm_small.hip (Hp <= 32): register-resident A fragments
    float* d_peep = nullptr;    // ocropy peephole cell: [ndir][3 (i, f, o)][Hp] peephole weights (lstm_big_kernel)
    void* d_wrecsmx = nullptr;  // ... split bf16 for its bf16x3 variant (plans whose arithmetic is split-bf16)
    bool rec_x3 = false;        // run the recurrence on the bf16 cores and emit split planes
    bool opad = false;          // ... with every direction's units padded to Hp (hidden % 8 != 0): the stores stay 16-byte pieces, the cluster kernel takes the layer; the consumer's weights get zero columns at the pad units (which are exactly 0: zero weights, zero bias)
    bool on_split = false;      // MAXPOOL / GN / TOSEQ working on split-bf16 NHWC planes (norm_x3.hip)
    bool split_rows = false;    // TOSEQ: fp32 NCHW in, K-blocked split sequence rows out (toseq_split_f32)
    int img_axis = 0;           // LSTM over image rows (1) or columns (2): sequences = N*H (N*W), steps = W (H); 0 = plain sequence
    int yaxis = 0;              // IMG2ROWS / ROWS2IMG: 1 = columns are the sequences
    // parallel groups (MultiParamParallel): S_ALIAS makes the output of step `src` (-1: the plan's input) the current tensor
    // again (a group member's input; no kernel); S_CONCAT joins the members' outputs `srcs` = (step, channels, length stage)
    // on the channel axis; S_ADD sums the `nk` pieces of `chunk` entries of the channel (add_axis 0) or height (1) axis
    int src = -1;
    struct Member { int step, C, stage; };
    std::vector<Member> srcs;
    int add_axis = 0, chunk = 0, nk = 0;   // (add_axis 2: the width, 3: the batch -- the pieces are counted per call)
    int reshape = -1;           // S_PERMUTE: index into krk_plan::reshapes; in_seq: the input arrives as sequence rows (N, T, C)
    bool in_seq = false;
    int last_only = 0;          // ROWS2IMG: keep the last step of every column (summarising LSTM): output height 1; LSTM step: time steps of the rows
    // output description
    bool out_is_seq = false;
    int outC = 0, outH = 1;     // NCHW: channels,height; seq: features,1
    int len_in = 0, len_out = 0;  // indices into the per-stage length table
    // per-call
    DevBuf out, aux, aux2;
    bool in_split = false;    // input arrives as split bf16 planes (bf16x3 mode)
    // split-bf16 LSTM -> LSTM / linear: the rows between them stay tile-time-major (16-line tiles, ceil(N/16)*16*T rows):
    // the recurrent kernel writes whole 256-byte runs and gemm_x3 needs no permutation (the linear layer undoes it)
    bool out_tiled = false, in_tiled = false;
    int seq_kpad = 0;         // > 0: this step's split sequence rows feed a projection whose K is padded to seq_kpad: the planes are that wide, the last octet zeroed
    double flops = 0.0;
};

}  // namespace

struct krk_plan {
    int device = 0;
    int in_c = 1, in_h = 1;
    int precision = KRK_PREC_F32;
    std::vector<Step> steps;
    int nstages = 1;  // length-table rows: 0 = input widths
    // how lengths evolve: stage s+1 = f(stage s) for the steps that change the width
    // kind 0: conv (clamp min 1), 1: pool, 2: one column (L?xs), 3: zero insertion of a transposed convolution ((L - 1) s + 1),
    // 4: k columns (Addition over the width), 5: general Reshape number k of `reshapes` (lines AND width of the batch may change),
    // 6: k lines (Addition over the batch)
    struct LenOp { int kind; int k, s, d, p; int from; };
    // Reshape (reference layers.py:285-335), axes in NCHW numbering: axis `src` is split into a x b (one of them may be -1), the part
    // that is not kept moves in front of axis `high` / `low` and merges with it.  C, H: the static dims in front of the layer
    struct ReshapeOp { int src, a, b, high, low, C, H; };
    std::vector<ReshapeOp> reshapes;
    bool batch_ops = false;                        // some layer changes the number of lines (Addition / Reshape on the batch axis)
    std::vector<LenOp> lenops;                     // lenops[i] produces stage i+1 from stage `from` <= i (a tree: parallel groups)
    int out_stage = 0;                             // the stage of the plan's output
    DevBuf d_lens;
    int* h_lens_pinned = nullptr;
    size_t h_lens_cap = 0;
    hipEvent_t lens_ev = nullptr;
    bool lens_ev_pending = false;
    // cross-plan gating of the convolution block (krk_plan_front_event / krk_plan_wait_front)
    hipEvent_t front_ev = nullptr;      // recorded when this plan's convolution block has been enqueued-and-run
    hipEvent_t front_wait = nullptr;    // not owned: another plan's front_ev the next call waits for (one shot)
    DevBuf d_labels, d_confs, d_final;
    // device-side failure word (mapped host memory): set by a kernel that gave up waiting (lstm_ws.hip exchange timeout)
    unsigned* err_host = nullptr;
    unsigned* err_dev = nullptr;
    // the packed weights (every d_w* / d_gamma / ... pointer of `steps`) belong to this set, shared by the plan krk_plan_create built
    // and its krk_plan_clone copies; freed when the last of them is destroyed.  Null only while a plan is being built.
    std::shared_ptr<std::vector<void*>> weights;
    int recurrence = KRK_RECURRENCE_AUTO;   // krk_plan_set_recurrence: which recurrent kernel the split-bf16 layers of THIS plan take
    bool profiling = false;
    std::vector<hipEvent_t> events;          // one per profiled launch + 1
    std::vector<const char*> prof_names;     // kernel group of each profiled launch of the last call
    std::vector<double> prof_flops;
    size_t prof_n = 0;
    int last_N = 0, last_W = 0;
};

namespace {

// Reshape.forward (reference layers.py:313-333) on shapes: `in` = (N, C, H, W) -> the 5-D view `d5`, the reference's rotation of the
// permutation list `perm`, and the merged 4-D result `out`.  False when the parts do not divide the axis (torch's reshape raises).
bool reshape_dims(const krk_plan::ReshapeOp& r, const int in[4], int d5[5], int perm[5], int out[4], int* dest_out = nullptr) {
    int a = r.a, b = r.b;
    const int size = in[r.src];
    if (a == -1) { if (b <= 0 || size % b) return false; a = size / b; }
    else if (b == -1) { if (a <= 0 || size % a) return false; b = size / a; }
    if (a <= 0 || b <= 0 || (long)a * b != size) return false;
    for (int i = 0, j = 0; i < 4; ++i) {
        if (i == r.src) { d5[j++] = a; d5[j++] = b; }
        else d5[j++] = in[i];
    }
    // "dest = low; if high != src_dim: dest = high; else: src_dim += 1", then the element at src_dim is swapped step by step to dest
    int dest = r.low, sd = r.src;
    if (r.high != r.src) dest = r.high; else sd += 1;
    // ... i.e. position sd of the 5-D view travels to position dest and the positions in between close the gap: a rotation
    for (int i = 0; i < 5; ++i) perm[i] = i;
    if (dest > sd) std::rotate(perm + sd, perm + sd + 1, perm + dest + 1);
    else std::rotate(perm + dest, perm + sd, perm + sd + 1);
    int pd[5];
    for (int i = 0; i < 5; ++i) pd[i] = d5[perm[i]];
    for (int i = 0, j = 0; i < 5; ++i) {
        if (i == dest) { out[j++] = pd[i] * pd[i + 1]; ++i; }
        else out[j++] = pd[i];
    }
    if (dest_out) *dest_out = dest;
    return true;
}

// valid width of a line behind a length-changing layer; Win / Wout: the BATCH's widths in front of / behind it (Reshape only)
int width_after(const krk_plan& p, const krk_plan::LenOp& op, int L, int Win, int Wout) {
    if (op.kind == 2) return std::min(L, 1);
    if (op.kind == 3) return L > 0 ? (L - 1) * op.s + 1 : 0;
    if (op.kind == 4 || op.kind == 6) return L;   // Addition.forward hands the seq_lens through untouched (reference layers.py:205-210)
    // Reshape.forward, layers.py:331-332: (seq_len * (float(initial_len) / o.shape[3])).int() -- an int tensor times a Python float
    // is a float32 product (the double ratio rounded to float32 first), .int() truncates
    if (op.kind == 5) return Wout > 0 ? (int)((float)L * (float)((double)Win / (double)Wout)) : 0;
    if (op.kind == 0) return std::max(conv_out(L, op.k, op.s, op.d, op.p), 1);
    return floordiv(L - (op.k - 1) - 1, op.s) + 1;
}

// lines and width of the batch at every length stage (not clamped: shapes follow torch's conv/pool arithmetic; stage s+1 derives
// from stage lenops[s].from <= s).  Returns the first stage that cannot be formed (a Reshape that does not divide, an Addition
// over more lines than there are) or 0
int stage_dims(const krk_plan& p, int N, int W, std::vector<int>& Ns, std::vector<int>& Ws) {
    const size_t n = p.lenops.size() + 1;
    Ns.assign(n, N);
    Ws.assign(n, 0);
    Ws[0] = W;
    for (size_t s = 0; s + 1 < n; ++s) {
        const krk_plan::LenOp& op = p.lenops[s];
        const int w = Ws[op.from];
        Ns[s + 1] = Ns[op.from];
        if (op.kind == 2) Ws[s + 1] = std::min(w, 1);
        else if (op.kind == 3) Ws[s + 1] = w > 0 ? (w - 1) * op.s + 1 : 0;
        else if (op.kind == 4) Ws[s + 1] = op.k;
        else if (op.kind == 5) {
            const krk_plan::ReshapeOp& r = p.reshapes[op.k];
            const int in[4] = {Ns[op.from], r.C, r.H, w};
            int d5[5], perm[5], out[4];
            if (!reshape_dims(r, in, d5, perm, out)) return (int)s + 1;
            Ns[s + 1] = out[0];
            Ws[s + 1] = out[3];
        } else if (op.kind == 6) {
            if (Ns[op.from] < op.k) return (int)s + 1;
            Ns[s + 1] = op.k;
            Ws[s + 1] = w;
        } else if (op.kind == 0) Ws[s + 1] = conv_out(w, op.k, op.s, op.d, op.p);
        else Ws[s + 1] = floordiv(w - (op.k - 1) - 1, op.s) + 1;
    }
    return 0;
}

// valid widths of one line at every stage, before the clamp to the tensor width
void line_widths(const krk_plan& p, int L, const std::vector<int>& Ws, std::vector<int>& v) {
    v.resize(p.lenops.size() + 1);
    v[0] = L;
    for (size_t s = 0; s + 1 < v.size(); ++s) {
        const krk_plan::LenOp& op = p.lenops[s];
        v[s + 1] = width_after(p, op, v[op.from], Ws[op.from], Ws[s + 1]);
    }
}

// kg_force > 0: the K-group size the kernel expects whatever the block count (lstm_big_kernel: 4)
void pack_lstm_recurrent(const Step& st, const float* const* whh, int M, std::vector<float>& pack, int kg_force = 0) {
    const int H = st.hidden, Hp = st.Hp, G = 4 * Hp;
    const int KPI = (M == 32) ? 2 : 4;
    const int NB = G / M;
    const int KG = kg_force > 0 ? kg_force : krk_lstm_kg(M, (NB + 3) / 4);
    const int KS = Hp / KPI, NG = (KS + KG - 1) / KG;
    pack.assign((size_t)st.ndir * NG * NB * 64 * KG, 0.f);
    for (int d = 0; d < st.ndir; ++d)
        for (int g = 0; g < NG; ++g)
            for (int b = 0; b < NB; ++b)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < KG; ++e) {
                        const int ks = g * KG + e;
                        const int k = KPI * ks + ((M == 32) ? (lane >> 5) : (lane >> 4));
                        const int col = b * M + (lane & (M - 1));
                        const int u = col >> 2, gt = col & 3;
                        if (u >= H || k >= H) continue;
                        pack[((((size_t)d * NG + g) * NB + b) * 64 + lane) * KG + e] =
                            whh[d][((size_t)gt * H + u) * H + k];
                    }
}

// lstm_small.hip: [dir][b][ks][lane] = W[gate column 16*b + (lane&15)][k = 4*ks + (lane>>4)], Hp/4 blocks and K steps
void pack_lstm_small(const Step& st, const float* const* whh, std::vector<float>& pack) {
    const int H = st.hidden, NB = st.Hp / 4;
    pack.assign((size_t)st.ndir * NB * NB * 64, 0.f);
    for (int d = 0; d < st.ndir; ++d)
        for (int b = 0; b < NB; ++b)
            for (int ks = 0; ks < NB; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int col = 16 * b + (lane & 15), k = 4 * ks + (lane >> 4);
                    const int u = col >> 2, gt = col & 3;
                    if (u >= H || k >= H) continue;
                    pack[(((size_t)d * NB + b) * NB + ks) * 64 + lane] = whh[d][((size_t)gt * H + u) * H + k];
                }
}

// [dir][kb][block][plane][lane][8]: lane l holds K = kb*32 + 8*(l>>4) + e of gate column block*16 + (l&15)
int upload_lstm_x3(Step& st, const float* const* whh) {
    const int H = st.hidden, Hp = st.Hp, G = 4 * Hp;
    const int NB = G / 16, NKB = (Hp + 31) / 32;
    std::vector<uint16_t> pack((size_t)st.ndir * NKB * NB * 1024, 0);
    for (int d = 0; d < st.ndir; ++d)
        for (int kb = 0; kb < NKB; ++kb)
            for (int b = 0; b < NB; ++b)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int k = kb * 32 + 8 * (lane >> 4) + e;
                        const int col = b * 16 + (lane & 15);
                        const int u = col >> 2, gt = col & 3;
                        if (u >= H || k >= H) continue;
                        const float v = whh[d][((size_t)gt * H + u) * H + k];
                        const uint16_t hi = f2bf(v);
                        const size_t base = (((size_t)d * NKB + kb) * NB + b) * 1024 + lane * 8 + e;
                        pack[base] = hi;
                        pack[base + 512] = f2bf(v - bf2f(hi));
                    }
    HIPCHK(hipMalloc(&st.d_wrecx3, pack.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(st.d_wrecx3, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    if (krk_lstm_ws_supported(st.opad ? Hp : H, Hp)) {
        // lstm_ws.hip: [dir][slice 4][wave 8][i][kb][plane][lane][8]; slice r owns blocks [r*BPC, (r+1)*BPC), wave w of it the
        // blocks r*BPC + w + 8i; fragments of blocks that do not exist stay zero (they compute h = 0 and publish nothing)
        const int BPC = (NB + 3) / 4, BPW = (BPC + 7) / 8;
        std::vector<uint16_t> pw((size_t)st.ndir * 32 * BPW * NKB * 1024, 0);
        for (int d = 0; d < st.ndir; ++d)
            for (int r = 0; r < 4; ++r)
                for (int w = 0; w < 8; ++w)
                    for (int i = 0; i < BPW; ++i) {
                        const int bl = w + 8 * i, b = r * BPC + bl;
                        if (bl >= BPC || b >= NB) continue;
                        for (int kb = 0; kb < NKB; ++kb)
                            std::memcpy(&pw[(((((size_t)d * 4 + r) * 8 + w) * BPW + i) * NKB + kb) * 1024],
                                        &pack[(((size_t)d * NKB + kb) * NB + b) * 1024], 1024 * sizeof(uint16_t));
                    }
        HIPCHK(hipMalloc(&st.d_wrecws, pw.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpy(st.d_wrecws, pw.data(), pw.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&st.ws_ctrl, 64));
        HIPCHK(hipMemset(st.ws_ctrl, 0, 64));
        st.ws_bpc = BPC;
    }
    return KRK_OK;
}

// lstm_small_x3_kernel: [dir][block b][plane][lane][8]: W[gate column 16 b + (lane & 15)][unit 4 j + (lane >> 4)], j = 0..7, split bf16
int upload_lstm_small_x3(Step& st, const float* const* whh) {
    const int H = st.hidden, NB = st.Hp / 4;
    std::vector<uint16_t> pack((size_t)st.ndir * NB * 2 * 64 * 8, 0);
    for (int d = 0; d < st.ndir; ++d)
        for (int b = 0; b < NB; ++b)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int col = 16 * b + (lane & 15), k = 4 * j + (lane >> 4);
                    const int u = col >> 2, gt = col & 3;
                    if (u >= H || k >= H) continue;
                    const float v = whh[d][((size_t)gt * H + u) * H + k];
                    const uint16_t hi = f2bf(v);
                    const size_t base = ((((size_t)d * NB + b) * 2) * 64 + lane) * 8 + j;
                    pack[base] = hi;
                    pack[base + 64 * 8] = f2bf(v - bf2f(hi));
                }
    HIPCHK(hipMalloc(&st.d_wrecsmx, pack.size() * sizeof(uint16_t)));
    HIPCHK(hipMemcpy(st.d_wrecsmx, pack.data(), pack.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return KRK_OK;
}

int upload(float** dst, const std::vector<float>& v) {
    HIPCHK(hipMalloc((void**)dst, std::max<size_t>(v.size(), 1) * sizeof(float)));
    if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return KRK_OK;
}

// every pointer of a step into the packed weights (krk_plan::weights)
std::vector<void*> step_weights(const Step& s) {
    return {s.cg.d_w, s.cg.d_b, s.cg.d_wx3, s.cg.d_wx5, s.cg.d_wx6, s.d_c1w, s.d_c1b, s.d_wrecsm, s.d_peep, s.d_wrecsmx, s.d_gamma,
            s.d_beta, s.d_wrec32, s.d_wrec16, s.d_wrecx3, s.d_wrecws};
}

// `weights`: false for a plan whose weights are owned by krk_plan::weights (shared with its clones)
void free_step(Step& s, bool weights) {
    if (weights)
        for (void* q : step_weights(s))
            if (q) (void)hipFree(q);
    if (s.ws_ctrl) (void)hipFree(s.ws_ctrl);
    s.ws_gran.release();
    s.out.release();
    s.aux.release();
    s.aux2.release();
}

// A step under construction owns what it has uploaded until it joins the plan: a layer the plan's arithmetic refuses half-way (the
// caller then falls back to the exact-f32 plan) must not leave its packed weights behind (found by the host-ASan harness, round 6:
// two allocations per refused plan).  `keep` is set right in front of the push_back.
struct StepGuard {
    Step& s;
    bool keep = false;
    ~StepGuard() { if (!keep) free_step(s, true); }
};

bool monotone_act(int act) { return act >= 0 && act <= KRK_ACT_SIGMOID; }

int map_act(int act) { return act == KRK_ACT_SIGMOID ? ACT_LINEAR : act; }

}  // namespace

extern "C" {

int krk_abi_version(void) { return KRK_ABI_VERSION; }

const char* krk_last_error(void) { return g_err.c_str(); }

int krk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void krk_plan_destroy(krk_plan* plan) {
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    (void)hipDeviceSynchronize();
    for (auto& s : plan->steps) free_step(s, plan->weights == nullptr);
    plan->weights.reset();                      // the last plan that shares them frees the packed weights
    plan->d_lens.release();
    plan->d_labels.release();
    plan->d_confs.release();
    plan->d_final.release();
    if (plan->h_lens_pinned) (void)hipHostFree(plan->h_lens_pinned);
    if (plan->err_host) (void)hipHostFree(plan->err_host);
    if (plan->lens_ev) (void)hipEventDestroy(plan->lens_ev);
    if (plan->front_ev) (void)hipEventDestroy(plan->front_ev);
    for (auto e : plan->events) (void)hipEventDestroy(e);
    delete plan;
}

}  // extern "C"

namespace {

// Compiles a layer table into a plan's schedule.  State that travels from layer to layer: the activation's shape and
// format (image / sequence, fp32 / split-bf16 planes), the length stage, and the lookahead index `i` (a convolution
// swallows a directly following 2x2 max-pool and the height->channel reshape).
constexpr int kRetryPlan = -100;   // PlanBuilder -> krk_plan_create: a zero-filter request was filed, compile again
constexpr int kNoPadPlan = -101;   // ... a padded activation met a layer that cannot take it: compile again without padding

struct PlanBuilder {
    krk_plan* p;
    const krk_layer* layers;
    int n_layers;
    bool x3;                  // split-bf16 kernels in use for the layer being compiled (see build(): off up to the last GroupNorm)
    int C, H;
    bool seq = false;
    bool split_fmt = false;   // bf16x3: the current activation is held as split bf16 planes
    // Round 6: a split-bf16 convolution needs input channels in blocks of 16.  A producer with 20 / 24 / 40 filters is compiled with ZERO
    // filters appended (weights and biases 0; whatever the activation makes of 0 meets zero weights in the consumer) when its consumer
    // asks for it: the request is filed under the producer's layer index and the plan compiled again (krk_plan_create's loop) -- the
    // rest of such a stack no longer runs on the exact-f32 kernels.
    std::map<int, int>* cpad = nullptr;   // layer index -> number of filters incl. the zero ones (survives the retries)
    int c_log = 0;                        // > 0: the current activation holds c_log real channels in front of zero ones (C = all of them)
    bool pad_conflict = false;            // a padded activation reached a layer that cannot take it: compile again without padding
    int last_split_conv = -1;             // layer index of the convolution that wrote the current split planes
    std::vector<int> seq_colmap;   // non-empty: the split sequence rows in front hold every direction's units padded to Hp (Step::opad): feature -> column
    int seq_kphys = 0;
    int stage = 0;
    int i = 0;
    bool want_x3 = false;     // the plan's arithmetic is split-bf16 (KRK_PREC_BF16X3 / KRK_PREC_BF16)
    bool left_x3 = false;     // a layer that exists in the f32 plan only was met: the rest of the network stays there
    int last_gn = -1;         // index of the network's last GroupNorm layer

    // a layer that changes the width: a new row of the length table, derived from the current one
    void new_stage(int kind, int k, int s_, int d, int pd) {
        p->lenops.push_back({kind, k, s_, d, pd, stage});
        stage = (int)p->lenops.size();
    }

    // parallel groups being compiled, innermost last
    struct Fork {
        int src, C, H, stage;
        bool seq;
        std::vector<Step::Member> members;
        int outH = 0;
        bool outseq = false;
    };
    std::vector<Fork> forks;
    int last_f32_only = -1;   // index of the last layer that exists in the exact-f32 arithmetic only (GroupNorm, group markers, A)

    int par_begin(const std::string& where);
    int par_member_done(const std::string& where);
    int par_next(const std::string& where);
    int par_end(const std::string& where);
    int addition(const krk_layer& L, const std::string& where);
    void push_alias(int src) {
        Step a;
        a.kind = S_ALIAS;
        a.src = src;
        a.C = C; a.H = H;
        a.out_is_seq = seq;
        a.outC = C; a.outH = H;
        a.len_in = a.len_out = stage;
        p->steps.push_back(std::move(a));
    }

    void push_toseq() {
        Step s;
        s.kind = S_TOSEQ;
        s.C = C;
        s.H = H;
        s.out_is_seq = true;
        s.outC = C * H;
        s.outH = 1;
        s.len_in = s.len_out = stage;
        p->steps.push_back(std::move(s));
        seq = true;
        C = C * H;
        H = 1;
    }

    // bf16x3 plans cover what the split-operand kernels implement; a layer that only exists in the f32 plan (an LSTM over
    // image rows/columns, GroupNorm on a channel count that is not a power of two, ...) does not reject the network:
    // the activations are converted once (split NHWC -> fp32 NCHW) and the rest of the plan runs on the f32 kernels.
    void leave_x3() {
        if (split_fmt && !seq) {
            Step u;
            u.kind = S_UNSPLIT;
            u.C = C; u.H = H;
            u.outC = C; u.outH = H;
            u.len_in = u.len_out = stage;
            p->steps.push_back(std::move(u));
        }
        if (c_log > 0) pad_conflict = true;   // fp32 NCHW with zero channels in it: nobody behind expects them
        x3 = false;
        left_x3 = true;
        split_fmt = false;
    }

    int conv(const krk_layer& L, const std::string& where, int ph_force = -1, int pw_force = -1);
    int conv_transposed(const krk_layer& L, const std::string& where);
    int maxpool(const krk_layer& L, const std::string& where);
    int groupnorm(const krk_layer& L, const std::string& where);
    int reshape(const krk_layer& L, const std::string& where);
    int reshape_general(const krk_layer& L, const std::string& where);
    int recurrent_or_linear(const krk_layer& L, const std::string& where);
    int linear(const krk_layer& L, const std::string& where, Step& s);
    int lstm(const krk_layer& L, const std::string& where, Step& s);
    int build();
};

// Can this (already planned) convolution hand its output to the tap kernel, i.e. write [N][H][C][pitch] planes?  conv1_x3.hip does
// (one grayscale channel in); so does the exact-f32 kernel when it is a plan's first split-bf16 layer -- RGB recognisers, first
// layers outside conv1_x3's geometry (round 4: their second convolution ran on the generic channel-as-K kernel, 1.8 ms per
// 256-line batch where the tap kernel takes 0.5).  KRK_NO_F32_NHCW keeps the round-3 routing.
bool feeds_taps(const ConvGeom& g) {
    if (g.c1x3) return g.Cout <= 32;          // (a first layer of more than 32 filters is two launches on channels-last planes)
    return g.split_out && !g.x3 && !g.taps && !g.x6 && !g.out_seq && !getenv("KRK_NO_F32_NHCW");
}

// ActConv2D (reference layers.py:791-860), with a directly following 2x2/2 MaxPool and/or the S reshape fused in
int PlanBuilder::conv(const krk_layer& L, const std::string& where, int ph_force, int pw_force) {
    if (seq) return fail(KRK_E_UNSUPPORTED, where + ": convolution after a sequence layer");
    if (!L.w[0] || !L.w[1]) return fail(KRK_E_INVALID, where + ": conv weights missing");
    if (L.cout <= 0 || L.kh <= 0 || L.kw <= 0 || L.sh <= 0 || L.sw <= 0 || L.dh <= 0 || L.dw <= 0)
        return fail(KRK_E_INVALID, where + ": bad conv geometry");
    if (L.act < 0 || L.act > KRK_ACT_SOFTMAX) return fail(KRK_E_UNSUPPORTED, where + ": activation");
    // Softmax over the channels (Cm..., O2s...): a linear convolution and a pass over its fp32 NCHW output; nothing fuses across it
    const bool softmax = L.act == KRK_ACT_SOFTMAX;
    if (softmax && (x3 || split_fmt)) return fail(KRK_E_UNSUPPORTED, where + ": channel-softmax convolution on split-bf16 planes");
    if (x3 && split_fmt && C % 16) {   // conv_x3 wants 16-channel K blocks; the tap kernel takes multiples of 4 after conv1_x3
        const bool taps_ok = !p->steps.empty() && p->steps.back().kind == S_CONV && feeds_taps(p->steps.back().cg) &&
                             krk_conv_taps_supported(C, L.cout, L.kh, L.kw, L.sh, L.sw, L.dh, L.dw) &&
                             !(i + 1 < n_layers && layers[i + 1].op == KRK_OP_RESHAPE_HC);
        if (!taps_ok) {
            if (cpad && c_log == 0 && last_split_conv >= 0 && !cpad->count(last_split_conv)) {
                (*cpad)[last_split_conv] = (C + 15) / 16 * 16;      // the producer again, with zero filters: compile the plan anew
                return kRetryPlan;
            }
            leave_x3();
        }
    }
    const int li = i;                                   // (i moves on when a pool / the reshape is fused below)
    const int Cin_log = c_log > 0 ? c_log : C;          // input channels of the torch weights
    int Cout_eff = L.cout;
    if (cpad && x3) {
        const auto it = cpad->find(li);
        if (it != cpad->end()) Cout_eff = it->second;
    }
    std::vector<float> wpad, bpad;
    const float* w_eff = L.w[0];
    const float* b_eff = L.w[1];
    if (Cout_eff != L.cout || Cin_log != C) {
        const size_t kk = (size_t)L.kh * L.kw;
        wpad.assign((size_t)Cout_eff * C * kk, 0.f);
        bpad.assign((size_t)Cout_eff, 0.f);
        for (int f = 0; f < L.cout; ++f) {
            for (int c = 0; c < Cin_log; ++c)
                std::memcpy(&wpad[((size_t)f * C + c) * kk], L.w[0] + ((size_t)f * Cin_log + c) * kk, kk * sizeof(float));
            bpad[f] = L.w[1][f];
        }
        w_eff = wpad.data();
        b_eff = bpad.data();
        if (!x3) pad_conflict = true;                   // (an exact-f32 layer behind a padded one: not planned for)
    }
    Step s;
    StepGuard guard{s};
    s.kind = S_CONV;
    s.C = C;
    s.H = H;
    ConvGeom& g = s.cg;
    g.Cin = C;
    g.H = H;
    g.Cout = Cout_eff;
    g.kh = L.kh; g.kw = L.kw; g.sh = L.sh; g.sw = L.sw; g.dh = L.dh; g.dw = L.dw;
    g.ph = ph_force >= 0 ? ph_force : (L.dh * (L.kh - 1)) / 2;
    g.pw = pw_force >= 0 ? pw_force : (L.dw * (L.kw - 1)) / 2;
    g.act = softmax ? ACT_LINEAR : map_act(L.act);
    s.len_in = stage;
    new_stage(0, L.kw, L.sw, L.dw, g.pw);
    const int Ho = conv_out(H, L.kh, L.sh, L.dh, g.ph);
    if (Ho <= 0) return fail(KRK_E_INVALID, where + ": conv output height <= 0");
    // fuse a directly following 2x2/2 max-pool, or the height->channel reshape
    if (i + 1 < n_layers && layers[i + 1].op == KRK_OP_MAXPOOL && layers[i + 1].kh == 2 &&
        layers[i + 1].kw == 2 && layers[i + 1].sh == 2 && layers[i + 1].sw == 2 && Ho >= 2 &&
        monotone_act(L.act)) {
        g.pool = true;
        new_stage(1, 2, 2, 1, 0);
        ++i;
        // bf16x3: conv_x3.hip can pool AND write the collapsed sequence rows in one epilogue
        if (x3 && split_fmt && i + 1 < n_layers && layers[i + 1].op == KRK_OP_RESHAPE_HC) {
            g.out_seq = true;
            ++i;
        }
    } else if (!softmax && i + 1 < n_layers && layers[i + 1].op == KRK_OP_RESHAPE_HC) {
        g.out_seq = true;
        ++i;
    }
    plan_conv_geom(g);
    if (upload_conv_weights(g, w_eff, b_eff, nullptr, nullptr) != KRK_OK) return KRK_E_HIP;
    // A convolution of a split-bf16 plan that must keep fp32-class operands (it feeds a GroupNorm, directly or through later
    // layers: i <= last_gn) takes the three-plane kernel (conv_x6.hip) when its geometry fits: fp32 NCHW in and out like the f32
    // kernel it replaces, 6/16 of its matrix time.  KRK_NO_CONV_X6 keeps the exact-f32 kernel.
    if (want_x3 && !left_x3 && !x3 && !g.out_seq && g.Cin % 16 == 0 && !getenv("KRK_NO_CONV_X6") && plan_x6_geom(g) == 0) {
        if (upload_x6_weights(g, w_eff) != KRK_OK) return KRK_E_HIP;
        g.x6 = true;
    }
    if (x3) {
        // the first convolution reads the caller's fp32 NCHW image on the f32 cores and hands
        // over split channels-last planes; every later one runs on the bf16 cores
        const bool first = !split_fmt;
        const int feat = g.out_seq ? g.Hy * Cout_eff : Cout_eff;
        if (g.out_seq && Cout_eff != L.cout) pad_conflict = true;      // (nobody asks a collapsing convolution for zero filters)
        if (feat % 4) return fail(KRK_E_UNSUPPORTED, where + ": bf16x3 needs a multiple of 4 output channels");   // the consumer checks its own K granule
        if (first && g.out_seq) return fail(KRK_E_UNSUPPORTED, where + ": bf16x3 needs >= 2 convolutions before the reshape");
        g.split_out = true;
        s.in_split = !first;
        if (!first && !g.out_seq && !p->steps.empty() && p->steps.back().kind == S_CONV && feeds_taps(p->steps.back().cg) &&
            krk_conv_taps_supported(g.Cin, g.Cout, g.kh, g.kw, g.sh, g.sw, g.dh, g.dw) && !getenv("KRK_NO_CONV_TAPS")) {
            g.taps = true;
            p->steps.back().cg.out_nhcw = true;
            if (upload_conv_taps_weights(g, w_eff) != KRK_OK) return KRK_E_HIP;
        } else if (!first) {
            g.x3 = true;
            if (plan_x3_geom(g) != KRK_OK || upload_x3_weights(g, w_eff, nullptr) != KRK_OK) return KRK_E_UNSUPPORTED;
        } else if (!g.out_seq && krk_conv1_x3_supported(g.Cin, g.Cout, g.kh, g.kw, g.sh, g.sw, g.dh, g.dw) &&
                   !getenv("KRK_NO_CONV1_X3")) {
            g.c1x3 = true;
            if (upload_conv1_x3_weights(g, w_eff) != KRK_OK) return KRK_E_HIP;
        }
        split_fmt = true;
        last_split_conv = li;
    }
    c_log = (Cout_eff != L.cout) ? L.cout : 0;
    s.len_out = stage;
    if (g.out_seq) {
        s.out_is_seq = true;
        s.outC = g.Hy * g.Cout;   // Hy == Ho unless a pool is fused in front of the reshape
        s.outH = 1;
        seq = true;
        C = s.outC;
        H = 1;
    } else {
        s.outC = g.Cout;
        s.outH = g.Hy;
        C = g.Cout;
        H = g.Hy;
    }
    guard.keep = true;
    p->steps.push_back(std::move(s));
    if (softmax) {
        Step m;
        m.kind = S_SOFTMAXC;
        m.C = C; m.H = H;
        m.outC = C; m.outH = H;
        m.len_in = m.len_out = stage;
        p->steps.push_back(std::move(m));
    }
    return KRK_OK;
}

// stand-alone MaxPool (reference layers.py:381-388): the ones a convolution could not swallow
int PlanBuilder::maxpool(const krk_layer& L, const std::string& where) {
    if (x3 && split_fmt && !seq && C % 8) leave_x3();
    const bool x3 = this->x3 && split_fmt;     // nothing split yet (pool in front of the first split-bf16 layer): the f32 kernel
    if (seq) return fail(KRK_E_UNSUPPORTED, where + ": max-pool after a sequence layer");
    if (L.kh <= 0 || L.kw <= 0 || L.sh <= 0 || L.sw <= 0) return fail(KRK_E_INVALID, where + ": bad pool");
    Step s;
    s.kind = S_MAXPOOL;
    s.C = C;
    s.H = H;
    s.kh = L.kh; s.kw = L.kw; s.sh = L.sh; s.sw = L.sw;
    s.on_split = x3;
    s.Ho = floordiv(H - (L.kh - 1) - 1, L.sh) + 1;
    if (s.Ho <= 0) return fail(KRK_E_INVALID, where + ": pool output height <= 0");
    s.len_in = stage;
    new_stage(1, L.kw, L.sw, 1, 0);
    s.len_out = stage;
    s.outC = C;
    s.outH = s.Ho;
    H = s.Ho;
    p->steps.push_back(std::move(s));
    return KRK_OK;
}

// GroupNorm (reference layers.py:967-984)
int PlanBuilder::groupnorm(const krk_layer& L, const std::string& where) {
    // (never on split planes: build() keeps every layer up to the network's last GroupNorm on the f32 kernels)
    if (seq) return fail(KRK_E_UNSUPPORTED, where + ": group norm after a sequence layer");
    if (L.cout <= 0 || C % L.cout) return fail(KRK_E_INVALID, where + ": groups must divide channels");
    if (!L.w[0] || !L.w[1]) return fail(KRK_E_INVALID, where + ": group norm weights missing");
    const int gn_at = i;          // (i moves on when the MaxPool behind is taken in)
    Step s;
    StepGuard guard{s};
    s.kind = S_GN;
    s.C = C;
    s.H = H;
    s.groups = L.cout;
    s.on_split = false;
    std::vector<float> ga(L.w[0], L.w[0] + C), be(L.w[1], L.w[1] + C);
    if (upload(&s.d_gamma, ga) != KRK_OK || upload(&s.d_beta, be) != KRK_OK) return KRK_E_HIP;
    s.len_in = s.len_out = stage;
    s.outC = C;
    s.outH = H;
    // a directly following MaxPool is taken in the apply pass: the normalised full-size tensor is never written
    if (i + 1 < n_layers && layers[i + 1].op == KRK_OP_MAXPOOL && !getenv("KRK_NO_GN_POOL")) {
        const krk_layer& P = layers[i + 1];
        if (P.kh <= 0 || P.kw <= 0 || P.sh <= 0 || P.sw <= 0) return fail(KRK_E_INVALID, where + ": bad pool");
        const int Ho = floordiv(H - (P.kh - 1) - 1, P.sh) + 1;
        if (Ho <= 0) return fail(KRK_E_INVALID, where + ": pool output height <= 0");
        s.pooled = true;
        s.kh = P.kh; s.kw = P.kw; s.sh = P.sh; s.sw = P.sw;
        s.Ho = Ho;
        new_stage(1, P.kw, P.sw, 1, 0);
        ++i;
        s.len_out = stage;
        s.outH = Ho;
        H = Ho;
    }
    // A one-channel 3x3 first convolution directly in front: recomputed inside the two GroupNorm passes (c1gn.hip), its full-size
    // output is never written.  KRK_NO_C1GN keeps the three-kernel path (A/B probing, the tests' reference arithmetic).
    if (gn_at >= 1 && layers[gn_at - 1].op == KRK_OP_CONV && p->steps.size() == 1 && p->steps.back().kind == S_CONV && !getenv("KRK_NO_C1GN")) {
        Step& cs = p->steps.back();
        const ConvGeom& cgm = cs.cg;
        const krk_layer& CL = layers[gn_at - 1];
        // the MaxPool behind the GroupNorm decides with its GEOMETRY (2 x 2 / 2 or none), whether or not it is taken into the apply
        // pass (KRK_NO_GN_POOL): both forms of a network then run the same convolution arithmetic and stay bit-identical
        const bool next_pool = gn_at + 1 < n_layers && layers[gn_at + 1].op == KRK_OP_MAXPOOL;
        const krk_layer* NP = next_pool ? &layers[gn_at + 1] : nullptr;
        if (!cgm.pool && !cgm.out_seq && !cgm.x3 && !cgm.c1x3 && !cgm.split_out &&
            krk_c1gn_supported(cgm.Cin, cgm.Cout, cgm.kh, cgm.kw, cgm.sh, cgm.sw, cgm.dh, cgm.dw, s.groups, NP ? NP->kh : 0, NP ? NP->kw : 0,
                               NP ? NP->sh : 0, NP ? NP->sw : 0)) {
            std::vector<float> cw(CL.w[0], CL.w[0] + (size_t)cgm.Cout * 9), cb(CL.w[1], CL.w[1] + cgm.Cout);
            if (upload(&s.d_c1w, cw) != KRK_OK || upload(&s.d_c1b, cb) != KRK_OK) return KRK_E_HIP;
            s.c1gn = true;
            s.c1_act = cgm.act;
            cs.skip = true;
        }
    }
    guard.keep = true;
    p->steps.push_back(std::move(s));
    return KRK_OK;
}

// S1(1x0)1,3: height folded into channels (reference layers.py:313-335) when no convolution fused it
int PlanBuilder::reshape(const krk_layer& L, const std::string& where) {
    (void)L;
    if (x3 && split_fmt && !seq && C % 8) leave_x3();
    const bool x3 = this->x3 && split_fmt;
    if (seq) return fail(KRK_E_UNSUPPORTED, where + ": reshape after a sequence layer");
    const int Cimg = C;
    push_toseq();
    p->steps.back().on_split = x3;   // writes the K-blocked split sequence rows gemm_x3.hip reads
    // An exact-f32 image part (GroupNorm networks) in front of split-bf16 sequence layers: collapse AND split in one pass
    // (norm_x3.hip toseq_split_f32) instead of to_seq + split_rows -- one read and one write of the tensor instead of two each
    // (the kernel moves 8 channels per lane; the rows' K blocks want 16 features)
    if (this->x3 && !split_fmt && Cimg % 8 == 0 && C % 16 == 0 && i + 1 < n_layers &&
        (layers[i + 1].op == KRK_OP_LSTM || layers[i + 1].op == KRK_OP_LINEAR) && !getenv("KRK_NO_TOSEQ_SPLIT")) {
        p->steps.back().split_rows = true;
        split_fmt = true;
    }
    return KRK_OK;
}

// Reshape in general (reference layers.py:285-335; model.py:739-777): one axis is split in two, one part moves in front of another
// axis and merges with it -- a permuted copy of the fp32 tensor (permute5_kernel).  Channels and height behind it are static (the
// caller computed them like the reference's get_shape does, from the spec's input shape); lines and width follow the call.
int PlanBuilder::reshape_general(const krk_layer& L, const std::string& where) {
    if (split_fmt) return fail(KRK_E_UNSUPPORTED, where + ": reshape on split-bf16 planes");
    if (L.kh < 0 || L.kh > 3 || L.sw < 0 || L.sw > 3 || L.dh < 0 || L.dh > 3 || (L.sw != L.kh && L.dh != L.kh))
        return fail(KRK_E_INVALID, where + ": reshape: either high or low must be the source dimension");
    if ((L.kw < 1 && L.kw != -1) || (L.sh < 1 && L.sh != -1) || (L.kw == -1 && L.sh == -1))
        return fail(KRK_E_INVALID, where + ": reshape: part sizes");
    if (L.cout < 1 || L.dw < 1) return fail(KRK_E_INVALID, where + ": reshape: output channels / height missing");
    krk_plan::ReshapeOp r{L.kh, L.kw, L.sh, L.sw, L.dh, C, H};
    Step s;
    s.kind = S_PERMUTE;
    s.C = C; s.H = H;
    s.in_seq = seq;
    s.reshape = (int)p->reshapes.size();
    s.outC = L.cout; s.outH = L.dw;
    s.out_is_seq = false;
    s.len_in = stage;
    new_stage(5, s.reshape, 1, 1, 0);
    s.len_out = stage;
    if (L.kh == 0 || L.sw == 0 || L.dh == 0) p->batch_ops = true;
    p->reshapes.push_back(r);
    p->steps.push_back(std::move(s));
    C = L.cout; H = L.dw;
    seq = false;
    return KRK_OK;
}

// LinSoftmax's projection (reference layers.py:710-722)
int PlanBuilder::linear(const krk_layer& L, const std::string& where, Step& s) {
    ConvGeom& g = s.cg;
    if (L.cout <= 0 || !L.w[0] || !L.w[1]) return fail(KRK_E_INVALID, where + ": linear weights missing");
    s.kind = S_LINEAR;
    g.Cout = L.cout;
    plan_conv_geom(g);
    if (upload_conv_weights(g, L.w[0], L.w[1], nullptr, nullptr) != KRK_OK) return KRK_E_HIP;
    if (x3) {
        g.x3 = true;
        s.in_split = split_fmt;
        const bool padded = s.in_split && !seq_colmap.empty();
        if (upload_gemm_x3_weights(g, L.w[0], nullptr, padded ? &seq_colmap : nullptr, seq_kphys) != KRK_OK) return KRK_E_UNSUPPORTED;
        if (s.in_split && g.xK != g.Cin && !p->steps.empty()) p->steps.back().seq_kpad = g.xK;      // the producer leaves room for the pad octet
        seq_colmap.clear();
        split_fmt = false;   // fp32 rows out
    }
    s.outC = L.cout;
    C = L.cout;
    return KRK_OK;
}

// torch.nn.LSTM inside TransposedSummarizingRNN (reference layers.py:467-547): input projection as one GEMM over all
// steps + the recurrent weights in every kernel's fragment order
int PlanBuilder::lstm(const krk_layer& L, const std::string& where, Step& s) {
    ConvGeom& g = s.cg;
    s.kind = S_LSTM;
    s.hidden = L.cout;
    s.dirmode = L.direction;
    if (L.cout <= 0 || L.direction < 0 || L.direction > 2) return fail(KRK_E_INVALID, where + ": bad LSTM");
    s.ndir = (L.direction == KRK_DIR_BIDI) ? 2 : 1;
    for (int k = 0; k < 4 * s.ndir; ++k)
        if (!L.w[k]) return fail(KRK_E_INVALID, where + ": LSTM weights missing");
    s.Hp = (s.hidden + 7) / 8 * 8;
    // act = 1 on an LSTM layer: the legacy ocropy peephole cell (reference layers.py:72-186): w[4d + 3] holds the peephole vectors
    // (i, f, o: 3 x hidden) instead of a second bias; the generic-width f32 kernel runs it
    if (L.act != 0 && L.act != 1) return fail(KRK_E_INVALID, where + ": LSTM cell variant");
    const bool peep = L.act == 1;
    if (peep && s.ndir != 2) return fail(KRK_E_INVALID, where + ": the ocropy peephole cell is bidirectional");
    const bool big = s.Hp > 256 || peep;   // lstm_big_kernel (exact f32, generic width); the projections run in the plan's arithmetic
    const int H_ = s.hidden, G = 4 * s.Hp;
    g.Cout = s.ndir * G;
    plan_conv_geom(g);
    // packed projection column d*G + 4*u + gate  <-  torch row gate*H + u of direction d
    std::vector<float> wih((size_t)g.Cout * C, 0.f), bsum(g.Cout, 0.f);
    std::vector<int> rowmap(g.Cout, -1);
    for (int d = 0; d < s.ndir; ++d)
        for (int u = 0; u < H_; ++u)
            for (int gt = 0; gt < 4; ++gt) {
                const int col = d * G + 4 * u + gt, row = gt * H_ + u;
                rowmap[col] = col;  // identity on the staged matrix below
                std::memcpy(&wih[(size_t)col * C], L.w[4 * d + 0] + (size_t)row * C, (size_t)C * sizeof(float));
                bsum[col] = L.w[4 * d + 2][row] + (peep ? 0.f : L.w[4 * d + 3][row]);
            }
    if (upload_conv_weights(g, wih.data(), nullptr, &rowmap, &bsum) != KRK_OK) return KRK_E_HIP;
    if (x3) {
        g.x3 = true;
        s.in_split = split_fmt;
        const bool padded = s.in_split && !seq_colmap.empty();
        if (upload_gemm_x3_weights(g, wih.data(), &rowmap, padded ? &seq_colmap : nullptr, seq_kphys) != KRK_OK) return KRK_E_UNSUPPORTED;
        if (s.in_split && g.xK != g.Cin && !p->steps.empty()) p->steps.back().seq_kpad = g.xK;      // the producer leaves room for the pad octet
        seq_colmap.clear();
        // all but a final LSTM run the recurrence on the bf16 cores and hand over split planes
        // ... up to 256 hidden units on the cluster / streaming kernels, 257 ... 512 on the block-major streaming kernel (lstm_x3.hip,
        // round 6: they fell to the exact-f32 lstm_big_kernel, 30 ms per layer at 512); the peephole cell and wider layers stay there
        const bool big_x3 = big && !peep && s.Hp <= 512 && !getenv("KRK_NO_LSTM_X3B");
        s.rec_x3 = (i + 1 < n_layers) && (!big || big_x3);
        split_fmt = s.rec_x3;
        // hidden sizes that are not a multiple of 8 (kraken's classic Lbx100; 150, 75 ...): every direction is written Hp units wide
        s.opad = s.rec_x3 && (s.hidden % 8) != 0 && !getenv("KRK_NO_OPAD");
        if (s.opad) {
            seq_colmap.resize((size_t)s.ndir * s.hidden);
            for (int d = 0; d < s.ndir; ++d)
                for (int k = 0; k < s.hidden; ++k) seq_colmap[(size_t)d * s.hidden + k] = d * s.Hp + k;
            seq_kphys = s.ndir * s.Hp;
        }
    }
    const float* whh[2] = {L.w[1], s.ndir == 2 ? L.w[5] : nullptr};
    std::vector<float> pk;
    if (!big) {
        pack_lstm_recurrent(s, whh, 32, pk);
        if (upload(&s.d_wrec32, pk) != KRK_OK) return KRK_E_HIP;
    }
    pack_lstm_recurrent(s, whh, 16, pk, big ? 4 : 0);      // lstm_big_kernel reads K groups of 4 at any width
    if (upload(&s.d_wrec16, pk) != KRK_OK) return KRK_E_HIP;
    if (s.rec_x3 && upload_lstm_x3(s, whh) != KRK_OK) return KRK_E_HIP;
    if (peep) {
        std::vector<float> pv((size_t)s.ndir * 3 * s.Hp, 0.f);
        for (int d = 0; d < s.ndir; ++d)
            for (int k = 0; k < 3; ++k)
                std::memcpy(&pv[((size_t)d * 3 + k) * s.Hp], L.w[4 * d + 3] + (size_t)k * H_, (size_t)H_ * sizeof(float));
        if (upload(&s.d_peep, pv) != KRK_OK) return KRK_E_HIP;
    }
    if (!big && !s.rec_x3 && krk_lstm_small_supported(s.Hp) && !getenv("KRK_NO_LSTM_SMALL")) {
        pack_lstm_small(s, whh, pk);
        if (upload(&s.d_wrecsm, pk) != KRK_OK) return KRK_E_HIP;
        // a split-bf16 plan runs these small recurrences on the bf16 cores too (the 2-D LSTMs of the segmenter, a small final LSTM)
        if (want_x3 && !getenv("KRK_NO_LSTM_SMALL_X3") && upload_lstm_small_x3(s, whh) != KRK_OK) return KRK_E_HIP;
    }
    s.outC = s.ndir * s.hidden;
    C = s.outC;
    return KRK_OK;
}

int PlanBuilder::recurrent_or_linear(const krk_layer& L, const std::string& where) {
    // LSTM over the rows (kw = 0) or columns (kw = 1) of an image: reference TransposedSummarizingRNN on a
    // 4-D input (layers.py:519-547; the BLLA segmenter's Lbx/Lby pairs).  img2rows -> LSTM -> rows2img.
    const bool img_lstm = L.op == KRK_OP_LSTM && !seq && (H != 1 || L.kw == 1);
    // summarising: only the last step of every sequence is kept (layers.py:537-539).  Along the height (L?ys) the image loses
    // its rows, along the width (L?xs) its columns: the tensor is one column wide from there on
    const bool summarize = L.op == KRK_OP_LSTM && L.kh == 1;
    const bool sum_x = summarize && L.kw == 0;
    if (sum_x && (x3 || split_fmt)) return fail(KRK_E_UNSUPPORTED, where + ": x-axis summarising LSTM on split-bf16 planes");
    const int Himg = H;   // image height in front of the layer
    if (img_lstm) {
        if (x3) leave_x3();   // LSTMs over image rows/columns exist in the f32 plan only
        Step a;
        a.kind = S_IMG2ROWS;
        a.C = C; a.H = H; a.yaxis = L.kw == 1;
        a.outC = C; a.outH = H;
        a.len_in = a.len_out = stage;
        p->steps.push_back(std::move(a));
    }
    if (L.op == KRK_OP_LSTM && seq && L.kw == 1)
        return fail(KRK_E_UNSUPPORTED, where + ": y-axis LSTM after the height collapse");
    if (!seq && !img_lstm) {
        if (H != 1)
            return fail(KRK_E_UNSUPPORTED, where + ": recurrent/linear layer on an input of height " +
                                               std::to_string(H) + " (only height 1 is implemented)");
        if (x3 && split_fmt) return fail(KRK_E_UNSUPPORTED, where + ": bf16x3 needs the reshape fused into a convolution");
        push_toseq();
    }
    Step s;
    StepGuard guard{s};
    s.C = C;
    s.H = 1;
    s.out_is_seq = !img_lstm;
    s.outH = img_lstm ? H : 1;
    s.img_axis = img_lstm ? (L.kw == 1 ? 2 : 1) : 0;
    s.len_in = s.len_out = stage;
    ConvGeom& g = s.cg;
    g.Cin = C;
    g.H = 1;
    g.in_seq = g.out_seq = true;
    g.act = ACT_LINEAR;
    if (x3 && split_fmt && !img_lstm && !p->steps.empty() && p->steps.back().kind == S_LSTM && p->steps.back().rec_x3 &&
        !getenv("KRK_NO_TILED_ROWS")) {
        p->steps.back().out_tiled = true;
        s.in_tiled = true;
    }
    if (int rc = (L.op == KRK_OP_LINEAR) ? linear(L, where, s) : lstm(L, where, s)) return rc;
    guard.keep = true;
    p->steps.push_back(std::move(s));
    if (img_lstm || sum_x) {
        // sum_x on a plain sequence: rows (N, W, C) -> (N, 1, C), the same pick with one row per line
        Step b;
        b.kind = S_ROWS2IMG;
        b.C = C; b.H = Himg; b.yaxis = L.kw == 1;
        b.last_only = summarize;
        b.out_is_seq = !img_lstm;
        b.outC = C; b.outH = (summarize && !sum_x) ? 1 : Himg;
        b.len_in = stage;
        if (sum_x) new_stage(2, 1, 1, 1, 0);
        b.len_out = stage;
        p->steps.push_back(std::move(b));
        if (summarize && !sum_x) H = 1;
    }
    return KRK_OK;
}

// MultiParamParallel (reference layers.py:56-71, model.py:876-905).  Every member starts from the tensor in front of the group
// (an S_ALIAS step: no kernel, no copy -- every step owns its output buffer, nothing computes in place) and S_CONCAT gathers the
// members' outputs on the channel axis.
int PlanBuilder::par_begin(const std::string& where) {
    if (split_fmt) return fail(KRK_E_UNSUPPORTED, where + ": parallel group behind split-bf16 layers");
    Fork f;
    f.src = (int)p->steps.size() - 1;
    f.C = C; f.H = H; f.seq = seq; f.stage = stage;
    forks.push_back(f);
    push_alias(f.src);
    return KRK_OK;
}

int PlanBuilder::par_member_done(const std::string& where) {
    if (forks.empty()) return fail(KRK_E_INVALID, where + ": group marker outside a parallel group");
    Fork& f = forks.back();
    if (split_fmt) return fail(KRK_E_UNSUPPORTED, where + ": parallel member ends in split-bf16 planes");
    if (!f.members.empty() && (f.outH != H || f.outseq != seq))
        return fail(KRK_E_INVALID, where + ": Output shape in parallel block not equal!");
    f.outH = H;
    f.outseq = seq;
    f.members.push_back({(int)p->steps.size() - 1, C, stage});
    return KRK_OK;
}

int PlanBuilder::par_next(const std::string& where) {
    if (int rc = par_member_done(where)) return rc;
    const Fork& f = forks.back();
    C = f.C; H = f.H; seq = f.seq; stage = f.stage;
    push_alias(f.src);
    return KRK_OK;
}

int PlanBuilder::par_end(const std::string& where) {
    if (int rc = par_member_done(where)) return rc;
    Fork f = std::move(forks.back());
    forks.pop_back();
    Step c;
    c.kind = S_CONCAT;
    c.srcs = f.members;
    c.C = 0;
    for (const auto& m : f.members) c.C += m.C;
    c.H = f.outH;
    c.out_is_seq = f.outseq;
    c.outC = c.C; c.outH = f.outH;
    // valid widths behind the group: those of its LAST member (the reference keeps the seq_lens the last module returned)
    c.len_in = c.len_out = stage;
    p->steps.push_back(std::move(c));
    C = p->steps.back().C;
    H = f.outH;
    seq = f.outseq;
    return KRK_OK;
}

// Transposed convolution (ActConv2D(transposed=True), reference layers.py:826-834, model.py:701-712): ConvTranspose2d(stride s, padding
// p = d (k - 1) / 2, dilation d) == zeros inserted between the input's pixels (s - 1 per gap), then an ordinary convolution with the
// spatially flipped, (in, out)-swapped kernel, stride 1, dilation d, padding d (k - 1) - p.  The caller hands the weights over in that
// convolution's order (kraken_amd/vgsl.py); exact f32 (nothing in kraken's recognisers uses it: correctness, not speed).
int PlanBuilder::conv_transposed(const krk_layer& L, const std::string& where) {
    if (seq) return fail(KRK_E_UNSUPPORTED, where + ": convolution after a sequence layer");
    if (split_fmt) return fail(KRK_E_UNSUPPORTED, where + ": transposed convolution on split-bf16 planes");
    if (L.kh <= 0 || L.kw <= 0 || L.sh <= 0 || L.sw <= 0 || L.dh <= 0 || L.dw <= 0) return fail(KRK_E_INVALID, where + ": bad conv geometry");
    const int ph = (L.dh * (L.kh - 1)) / 2, pw = (L.dw * (L.kw - 1)) / 2;
    if (L.sh > 1 || L.sw > 1) {
        Step z;
        z.kind = S_UPZERO;
        z.C = C; z.H = H;
        z.sh = L.sh; z.sw = L.sw;
        z.outC = C;
        z.outH = H > 0 ? (H - 1) * L.sh + 1 : 0;
        z.len_in = stage;
        new_stage(3, 1, L.sw, 1, 0);
        z.len_out = stage;
        z.out_is_seq = false;
        H = z.outH;
        p->steps.push_back(std::move(z));
    }
    krk_layer C2 = L;
    C2.op = KRK_OP_CONV;
    C2.sh = C2.sw = 1;
    return conv(C2, where, L.dh * (L.kh - 1) - ph, L.dw * (L.kw - 1) - pw);
}

// Addition (reference layers.py:188-223): unfold(dim, chunk, chunk).sum(dim) -- out[j] = sum_k in[k*chunk + j]
int PlanBuilder::addition(const krk_layer& L, const std::string& where) {
    if (split_fmt) return fail(KRK_E_UNSUPPORTED, where + ": addition on split-bf16 planes");
    if (L.kh == 2) {     // over the width: the tensor is `chunk` columns wide from here on, the pieces are counted per call (W / chunk)
        if (seq) return fail(KRK_E_UNSUPPORTED, where + ": addition over the width of a sequence");
        if (L.cout < 1) return fail(KRK_E_INVALID, where + ": addition with chunk " + std::to_string(L.cout));
        Step a;
        a.kind = S_ADD;
        a.C = C; a.H = H;
        a.add_axis = 2; a.chunk = L.cout; a.nk = 0;
        a.outC = C; a.outH = H;
        a.len_in = stage;
        new_stage(4, L.cout, 1, 1, 0);
        a.len_out = stage;
        p->steps.push_back(std::move(a));
        return KRK_OK;
    }
    if (L.kh == 3) {     // over the batch: `chunk` lines from here on; the valid widths stay those of the call's lines (krk_layer doc)
        if (L.cout < 1) return fail(KRK_E_INVALID, where + ": addition with chunk " + std::to_string(L.cout));
        Step a;
        a.kind = S_ADD;
        a.C = C; a.H = H;
        a.add_axis = 3; a.chunk = L.cout; a.nk = 0;
        a.out_is_seq = seq;
        a.outC = C; a.outH = H;
        a.len_in = stage;
        new_stage(6, L.cout, 1, 1, 0);
        a.len_out = stage;
        p->batch_ops = true;
        p->steps.push_back(std::move(a));
        return KRK_OK;
    }
    const int size = L.kh == 0 ? C : H;
    if (L.kh < 0 || L.kh > 1 || L.cout < 1 || L.cout > size)
        return fail(KRK_E_INVALID, where + ": addition with chunk " + std::to_string(L.cout) + " on an axis of " + std::to_string(size));
    if (L.kh == 1 && seq) return fail(KRK_E_UNSUPPORTED, where + ": addition over the height of a sequence");
    Step a;
    a.kind = S_ADD;
    a.C = C; a.H = H;
    a.add_axis = L.kh; a.chunk = L.cout; a.nk = size / L.cout;
    a.out_is_seq = seq;
    a.outC = L.kh == 0 ? L.cout : C;
    a.outH = L.kh == 1 ? L.cout : H;
    a.len_in = a.len_out = stage;
    C = a.outC; H = a.outH;
    p->steps.push_back(std::move(a));
    return KRK_OK;
}

int PlanBuilder::build() {
    // Where the split-bf16 arithmetic may begin.  GroupNorm divides by the group's standard deviation and thereby amplifies the
    // error its input carries by |x| / sigma -- on random-weight networks rare lines of an all-split plan reached 2.3e-3 against
    // the 1e-3 parity gate (profiles/r02_fuzz_300s.txt); the error is that of the operands' 16-bit representation, so a fourth
    // (lo x lo) MFMA term would not remove it.  Every layer up to and including the LAST GroupNorm therefore runs on the exact-f32
    // matrix cores; the split-bf16 kernels take over behind it (the next convolution computes in f32 and hands over split planes;
    // sequence layers split their fp32 rows on the way in), where the error is the ~1e-5 of a GroupNorm-free network.
    want_x3 = x3;
    for (int k = 0; k < n_layers; ++k) {
        if (layers[k].op == KRK_OP_GROUPNORM) last_gn = k;
        // parallel groups and additions work on fp32 tensors: like the GroupNorm part, everything up to the last of them runs on
        // the exact-f32 kernels and the split-bf16 ones take over behind it
        if (layers[k].op == KRK_OP_GROUPNORM || (layers[k].op >= KRK_OP_PAR_BEGIN && layers[k].op <= KRK_OP_RESHAPE) ||
            (layers[k].op == KRK_OP_LSTM && layers[k].kh == 1 && layers[k].kw == 0) ||
            (layers[k].op == KRK_OP_CONV && layers[k].act == KRK_ACT_SOFTMAX))
            last_f32_only = k;
    }
    for (i = 0; i < n_layers; ++i) {
        const krk_layer& L = layers[i];
        const std::string where = "layer " + std::to_string(i);
        // (round 6: a convolution stack that left the split-bf16 kernels -- 24 / 40 / 48 channels: no 16-channel K blocks -- no longer takes
        // the sequence part with it: recurrent and linear layers over the width axis split their fp32 rows on the way in, like behind a
        // GroupNorm part; BENCH-A's layers on 20 / 40 channels ran their recurrences on the exact-f32 kernel, 2.9 ms each)
        const bool seq_layer = seq && (L.op == KRK_OP_LSTM || L.op == KRK_OP_LINEAR) && !(L.kh == 1 && L.kw == 0);
        x3 = want_x3 && i > last_f32_only && (!left_x3 || seq_layer);
        // a convolution that would be the first split-bf16 layer AND carry the height collapse has no split-plane hand-over
        // (the f32 kernel writes split NHWC planes, not sequence rows): it stays f32, the sequence layers behind it split their rows
        if (x3 && !split_fmt && !seq && L.op == KRK_OP_CONV) {
            int j = i + 1;
            if (j < n_layers && layers[j].op == KRK_OP_MAXPOOL) ++j;
            if (j < n_layers && layers[j].op == KRK_OP_RESHAPE_HC) x3 = false;
        }
        int rc;
        switch (L.op) {
            case KRK_OP_CONV: rc = conv(L, where); break;
            case KRK_OP_MAXPOOL: rc = maxpool(L, where); break;
            case KRK_OP_GROUPNORM: rc = groupnorm(L, where); break;
            case KRK_OP_RESHAPE_HC: rc = reshape(L, where); break;
            case KRK_OP_LSTM:
            case KRK_OP_LINEAR: rc = recurrent_or_linear(L, where); break;
            case KRK_OP_PAR_BEGIN: rc = par_begin(where); break;
            case KRK_OP_PAR_NEXT: rc = par_next(where); break;
            case KRK_OP_PAR_END: rc = par_end(where); break;
            case KRK_OP_ADD: rc = addition(L, where); break;
            case KRK_OP_CONVT: rc = conv_transposed(L, where); break;
            case KRK_OP_RESHAPE: rc = reshape_general(L, where); break;
            default: rc = fail(KRK_E_UNSUPPORTED, where + ": unknown op " + std::to_string(L.op));
        }
        if (rc) return rc;
        if (pad_conflict) return kNoPadPlan;
    }
    if (!forks.empty()) return fail(KRK_E_INVALID, "parallel group not closed");
    // a network that ENDS in a split-bf16 convolution (no sequence part): the caller gets fp32 NCHW like from every other plan
    if (split_fmt && !seq) leave_x3();
    if (pad_conflict) return kNoPadPlan;
    p->nstages = (int)p->lenops.size() + 1;
    p->out_stage = stage;
    return KRK_OK;
}

}  // namespace

namespace {

// what every plan has for itself, whether built or cloned: events, the mapped status word.  Returns what failed, or null.
const char* plan_private_state(krk_plan* p) {
    if (hipEventCreateWithFlags(&p->lens_ev, hipEventDisableTiming) != hipSuccess) return "hipEventCreate failed";
    if (hipEventCreateWithFlags(&p->front_ev, hipEventDisableTiming) != hipSuccess) return "hipEventCreate failed";
    if (hipHostMalloc((void**)&p->err_host, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&p->err_dev, p->err_host, 0) != hipSuccess)
        return "hipHostMalloc (status word) failed";
    *p->err_host = 0;
    return nullptr;
}

}  // namespace

extern "C" {

int krk_plan_clone(const krk_plan* src, krk_plan** out) {
    if (!src || !out) return fail(KRK_E_INVALID, "krk_plan_clone: null plan");
    if (!src->weights) return fail(KRK_E_INVALID, "krk_plan_clone: the source plan is not complete");
    HIPCHK(hipSetDevice(src->device));
    krk_plan* p = new krk_plan(*src);            // schedule, geometry, length rules and the weight POINTERS (the set is shared)
    // ... and nothing of the source's per-call state: workspaces grow on first use, events and the status word are its own
    for (auto& st : p->steps) {
        st.out = st.aux = st.aux2 = st.ws_gran = DevBuf{};
        st.ws_tickets = st.ws_epoch = 0;
        st.ws_ctrl = nullptr;
    }
    p->d_lens = p->d_labels = p->d_confs = p->d_final = DevBuf{};
    p->h_lens_pinned = nullptr;
    p->h_lens_cap = 0;
    p->lens_ev = p->front_ev = p->front_wait = nullptr;
    p->lens_ev_pending = false;
    p->err_host = p->err_dev = nullptr;
    p->profiling = false;
    p->events.clear();
    p->prof_names.clear();
    p->prof_flops.clear();
    p->prof_n = 0;
    p->last_N = p->last_W = 0;
    auto bail = [&](const std::string& msg) {
        krk_plan_destroy(p);
        return fail(KRK_E_HIP, msg);
    };
    for (size_t i = 0; i < p->steps.size(); ++i)
        if (src->steps[i].ws_ctrl) {             // the cluster kernel's ticket counter is written by every launch
            if (hipMalloc((void**)&p->steps[i].ws_ctrl, 64) != hipSuccess || hipMemset(p->steps[i].ws_ctrl, 0, 64) != hipSuccess)
                return bail("krk_plan_clone: hipMalloc (ticket counter) failed");
        }
    if (const char* what = plan_private_state(p)) return bail(what);
    HIPCHK(hipDeviceSynchronize());
    *out = p;
    return KRK_OK;
}

int krk_plan_create(const krk_layer* layers, int n_layers, int in_channels, int in_height, int precision,
                    int device, krk_plan** out) {
    if (!layers || n_layers <= 0 || !out) return fail(KRK_E_INVALID, "krk_plan_create: null/empty layer list");
    if (in_channels <= 0 || in_height <= 0)
        return fail(KRK_E_UNSUPPORTED, "krk_plan_create: input channels/height must be fixed and positive");
    if (precision != KRK_PREC_F32 && precision != KRK_PREC_BF16X3 && precision != KRK_PREC_BF16)
        return fail(KRK_E_UNSUPPORTED, "krk_plan_create: precision must be KRK_PREC_F32, KRK_PREC_BF16X3 or KRK_PREC_BF16");
    if (krk_device_count() <= device)
        return fail(KRK_E_HIP, "krk_plan_create: no HIP device " + std::to_string(device));
    HIPCHK(hipSetDevice(device));

    krk_plan* p = nullptr;
    auto bail = [&](int code, const std::string& msg) {
        krk_plan_destroy(p);
        return fail(code, msg);
    };
    // zero-filter requests (PlanBuilder::cpad): every retry compiles the whole plan again with one more producer padded; a padded plan
    // that fails for ANY reason is compiled once more without padding (what rounds 1-5 did)
    std::map<int, int> cpad;
    bool allow_pad = !getenv("KRK_NO_CPAD");
    for (int attempt = 0;; ++attempt) {
        p = new krk_plan();
        p->device = device;
        p->in_c = in_channels;
        p->in_h = in_height;
        p->precision = precision;
        // KRK_PREC_BF16 = the split-bf16 plan with the cross terms dropped (one MFMA per product): same layouts, same kernels
        PlanBuilder b{p, layers, n_layers, precision == KRK_PREC_BF16X3 || precision == KRK_PREC_BF16, in_channels, in_height};
        b.cpad = allow_pad ? &cpad : nullptr;
        const int rc = b.build();
        if (!rc) break;
        krk_plan_destroy(p);
        p = nullptr;
        if (rc == kRetryPlan && attempt < n_layers + 2) continue;
        if (allow_pad && !cpad.empty()) {          // (kNoPadPlan, or any failure of a padded plan)
            cpad.clear();
            allow_pad = false;
            continue;
        }
        return rc == kRetryPlan || rc == kNoPadPlan ? fail(KRK_E_UNSUPPORTED, "krk_plan_create: channel padding did not converge") : rc;
    }
    {
        const int dev = device;
        auto* set = new std::vector<void*>();
        for (const auto& st : p->steps)
            for (void* q : step_weights(st))
                if (q) set->push_back(q);
        p->weights = std::shared_ptr<std::vector<void*>>(set, [dev](std::vector<void*>* v) {
            int cur = 0;
            (void)hipGetDevice(&cur);
            (void)hipSetDevice(dev);
            for (void* q : *v) (void)hipFree(q);
            (void)hipSetDevice(cur);
            delete v;
        });
    }
    if (const char* what = plan_private_state(p)) return bail(KRK_E_HIP, what);
    HIPCHK(hipDeviceSynchronize());
    *out = p;
    return KRK_OK;
}

int krk_plan_out_dims(const krk_plan* plan, int N, int W, int* Nout, int* C, int* H, int* Wout) {
    if (!plan || plan->steps.empty()) return fail(KRK_E_INVALID, "krk_plan_out_dims: null plan");
    std::vector<int> Ns, Ws;
    if (const int bad = stage_dims(*plan, N, W, Ns, Ws))
        return fail(KRK_E_INVALID, "krk_plan_out_dims: a batch of " + std::to_string(N) + " lines of width " + std::to_string(W) +
                                   " does not fit the " + (plan->lenops[bad - 1].kind == 6 ? "addition" : "reshape") + " of length stage " +
                                   std::to_string(bad));
    const Step& last = plan->steps.back();
    if (Nout) *Nout = Ns[plan->out_stage];
    if (C) *C = last.outC;
    if (H) *H = last.outH;
    if (Wout) *Wout = Ws[plan->out_stage];
    return KRK_OK;
}

int krk_plan_out_shape(const krk_plan* plan, int W, int* C, int* H, int* Wout) {
    if (!plan || plan->steps.empty()) return fail(KRK_E_INVALID, "krk_plan_out_shape: null plan");
    if (plan->batch_ops) return fail(KRK_E_INVALID, "krk_plan_out_shape: the network changes the batch size: krk_plan_out_dims");
    return krk_plan_out_dims(plan, 1, W, nullptr, C, H, Wout);
}

int krk_plan_olens_w(const krk_plan* plan, const int* lens_host, int N, int W, int* olens_host) {
    if (!plan || !lens_host || !olens_host || N < 0) return fail(KRK_E_INVALID, "krk_plan_olens: bad argument");
    std::vector<int> Ns, Ws, v;
    if (plan->reshapes.empty() && W <= 0) {
        Ws.assign(plan->lenops.size() + 1, 0);         // no layer reads the batch's widths
    } else if (stage_dims(*plan, N, W, Ns, Ws)) {
        return fail(KRK_E_INVALID, "krk_plan_olens: a batch of " + std::to_string(N) + " lines of width " + std::to_string(W) +
                                   " does not fit this network");
    }
    for (int n = 0; n < N; ++n) {
        line_widths(*plan, lens_host[n], Ws, v);
        olens_host[n] = v[plan->out_stage];
    }
    return KRK_OK;
}

int krk_plan_olens(const krk_plan* plan, const int* lens_host, int N, int* olens_host) {
    if (plan && !plan->reshapes.empty())
        return fail(KRK_E_INVALID, "krk_plan_olens: a Reshape layer scales seq_lens by the batch's widths: krk_plan_olens_w");
    return krk_plan_olens_w(plan, lens_host, N, 0, olens_host);
}

long krk_plan_workspace_bytes(const krk_plan* plan) {
    if (!plan) return 0;
    size_t t = plan->d_lens.cap + plan->d_labels.cap + plan->d_confs.cap + plan->d_final.cap;
    for (const auto& s : plan->steps) t += s.out.cap + s.aux.cap;
    return (long)t;
}

int krk_plan_set_profiling(krk_plan* plan, int enable) {
    if (!plan) return fail(KRK_E_INVALID, "null plan");
    plan->profiling = enable != 0;
    if (plan->profiling && plan->events.empty()) {
        plan->events.resize(3 * plan->steps.size() + 1);
        for (auto& e : plan->events)
            if (hipEventCreate(&e) != hipSuccess) return fail(KRK_E_HIP, "hipEventCreate failed");
    }
    return KRK_OK;
}

int krk_plan_status(krk_plan* plan) {
    if (!plan) return fail(KRK_E_INVALID, "krk_plan_status: null plan");
    if (plan->err_host && *(volatile unsigned*)plan->err_host != 0) {
        const unsigned word = *(volatile unsigned*)plan->err_host;
        *(volatile unsigned*)plan->err_host = 0;
        char msg[256];
        snprintf(msg, sizeof msg, "a recurrent cluster kernel timed out waiting for its peers (lstm_ws exchange, word 0x%08x); the "
                                  "results of the batches in flight on this plan are invalid", word);
        return fail(KRK_E_HIP, msg);
    }
    return KRK_OK;
}

int krk_plan_has_exchange(const krk_plan* plan) {
    if (!plan || plan->recurrence == KRK_RECURRENCE_STREAMING) return 0;
    // lstm_ws.hip is the only kernel that waits for other workgroups; it takes the recurrent layers above 64 hidden units that run on
    // the bf16 cores (recurrence_x3) -- and any the KRK_LSTM_V probe switch forces onto it
    for (const auto& s : plan->steps)
        if (s.kind == S_LSTM && s.rec_x3 && s.d_wrecws && (s.Hp > 64 || getenv("KRK_LSTM_V"))) return 1;
    return 0;
}

int krk_plan_set_recurrence(krk_plan* plan, int variant) {
    if (!plan) return fail(KRK_E_INVALID, "krk_plan_set_recurrence: null plan");
    if (variant != KRK_RECURRENCE_AUTO && variant != KRK_RECURRENCE_STREAMING)
        return fail(KRK_E_INVALID, "krk_plan_set_recurrence: variant must be KRK_RECURRENCE_AUTO or KRK_RECURRENCE_STREAMING");
    plan->recurrence = variant;
    return KRK_OK;
}

int krk_plan_get_recurrence(const krk_plan* plan) {
    if (!plan) return fail(KRK_E_INVALID, "krk_plan_get_recurrence: null plan");
    return plan->recurrence;
}

void* krk_plan_front_event(krk_plan* plan) { return plan ? (void*)plan->front_ev : nullptr; }

int krk_plan_wait_front(krk_plan* plan, void* event) {
    if (!plan) return fail(KRK_E_INVALID, "krk_plan_wait_front: null plan");
    plan->front_wait = (hipEvent_t)event;
    return KRK_OK;
}

int krk_plan_num_steps(const krk_plan* plan) {
    if (!plan) return 0;
    return plan->prof_n ? (int)plan->prof_n : (int)plan->steps.size();
}

const char* krk_plan_layer_name(const krk_plan* plan, int i) {
    if (!plan || i < 0) return nullptr;
    if (plan->prof_n) return i < (int)plan->prof_n ? plan->prof_names[i] : nullptr;
    return i < (int)plan->steps.size() ? kStepNames[plan->steps[i].kind] : nullptr;
}

double krk_plan_layer_flops(const krk_plan* plan, int i) {
    if (!plan || i < 0) return 0.0;
    if (plan->prof_n) return i < (int)plan->prof_n ? plan->prof_flops[i] : 0.0;
    return i < (int)plan->steps.size() ? plan->steps[i].flops : 0.0;
}

int krk_plan_layer_ms(krk_plan* plan, float* ms_host, int cap) {
    if (!plan || !ms_host) return fail(KRK_E_INVALID, "bad argument");
    if (!plan->profiling || plan->events.empty() || !plan->prof_n) return fail(KRK_E_INVALID, "profiling not enabled");
    HIPCHK(hipEventSynchronize(plan->events[plan->prof_n]));
    const int n = std::min<int>(cap, (int)plan->prof_n);
    for (int i = 0; i < n; ++i) HIPCHK(hipEventElapsedTime(&ms_host[i], plan->events[i], plan->events[i + 1]));
    return n;
}

}  // extern "C"

namespace {

// ---- kernel argument blocks from a step's geometry ------------------------------------------------------------------
void fill_conv(const ConvGeom& g, ConvArgs& a, const float* xin, float* yout, int Nn, int Wn, const int* li, const int* lo) {
    a.x = xin; a.y = yout; a.wpack = g.d_w; a.bias = g.d_b;
    a.len_in = li; a.len_out = lo;
    a.N = Nn; a.Cin = g.Cin; a.H = g.H; a.W = Wn;
    a.Cout = g.Cout; a.CBpad = g.CBpad;
    a.kh = g.kh; a.kw = g.kw; a.sh = g.sh; a.sw = g.sw; a.dh = g.dh; a.dw = g.dw; a.ph = g.ph; a.pw = g.pw;
    a.Ho = g.Ho;
    a.Wo = g.in_seq ? Wn : conv_out(Wn, g.kw, g.sw, g.dw, g.pw);
    a.Hy = g.Hy;
    a.Wy = g.pool ? floordiv(a.Wo - 2, 2) + 1 : a.Wo;
    a.act = g.act;
    a.cchunk = g.cchunk; a.nchunks = g.nchunks; a.Kc = g.Kc;
    a.KSG = g.KSG; a.KSG_last = g.KSG_last; a.KSGpad = g.KSGpad; a.KS4 = g.KS4; a.vec4 = g.vec4;
    a.IH = g.IH; a.IW = g.IW; a.RS = g.RS; a.PS = g.PS; a.SR = g.SR;
    a.tiles_h = (g.Ho + g.TH - 1) / g.TH;
    a.tiles_w = (a.Wo + g.TW - 1) / g.TW;
    a.otab_floats = g.otab_floats;
    a.y_split = nullptr; a.y_plane = 0; a.y_sn = a.y_sr = a.y_sc = 0;
}

void fill_x3(const ConvGeom& g, X3Args& a, const void* xin, size_t x_plane, void* yout, int Nn, int Wn, const int* li,
             const int* lo, int dbg) {
    a.x = (const __bf16*)xin; a.x_plane = x_plane; a.y = yout;
    a.wpack = (const __bf16*)g.d_wx3; a.bias = g.d_b;
    a.len_in = li; a.len_out = lo;
    a.N = Nn; a.Cin = g.Cin; a.H = g.H; a.W = Wn;
    a.Cout = g.Cout; a.CBpad = g.CBpad;
    a.kh = g.kh; a.kw = g.kw; a.sh = g.sh; a.sw = g.sw; a.dh = g.dh; a.dw = g.dw; a.ph = g.ph; a.pw = g.pw;
    a.Ho = g.Ho;
    a.Wo = g.in_seq ? Wn : conv_out(Wn, g.kw, g.sw, g.dw, g.pw);
    a.Hy = g.Hy;
    a.Wy = g.pool ? floordiv(a.Wo - 2, 2) + 1 : a.Wo;
    a.act = g.act;
    a.cchunk = g.xchunk; a.nchunks = g.xnchunks; a.KB = g.xKB; a.KB_last = g.xKB_last;
    a.IH = g.IH; a.IW = g.IW; a.PSTR = g.xPSTR; a.lds_plane = g.xplane; a.SR = g.SR;
    a.tiles_h = (g.Ho + g.TH - 1) / g.TH;
    a.tiles_w = (a.Wo + g.TW - 1) / g.TW;
    a.y_plane = 0; a.y_sn = a.y_sr = a.y_sc = 0; a.y_blkM = 0; a.y_cols = 0; a.y_f32 = 0;
    a.dbg = dbg;
    a.tps = 0;
}

void fill_gemm(const ConvGeom& g, GemmX3Args& a, const void* xin, size_t x_plane, float* yout, int rows, int dbg) {
    a.x = (const __bf16*)xin; a.x_plane = x_plane;
    a.w = (const __bf16*)g.d_wx3; a.bias = g.d_b; a.y = yout;
    a.M = rows; a.K = g.xK ? g.xK : g.Cin; a.Cout = g.Cout;
    a.ncg = (g.Cout + 127) / 128; a.ntiles = (rows + 255) / 256;
    a.act = g.act;
    a.tileT = 0;
    a.nlines = 0;
    a.dbg = dbg;
    a.nbuf = env_int("KRK_GEMM_SPREAD", 1) ? 3 : 2;   // gemm_x3.hip reads nbuf == 2 as "copies in front of the MFMAs" (A/B probe)
}

// strides of a split channels-last output (NHWC, or sequence rows when the reshape is fused)
void split_strides(const ConvGeom& g, int Wo_, int Wy_, long& sn, long& sr, long& sc) {
    if (g.out_seq) { sn = (long)Wo_ * g.Ho * g.Cout; sc = (long)g.Ho * g.Cout; sr = g.Cout; }
    else { sn = (long)g.Hy * Wy_ * g.Cout; sr = (long)Wy_ * g.Cout; sc = g.Cout; }
}

// Probe switches (not API): looked up once per forward call -- the tools and a few tests flip them
// between calls on a live plan.
struct Probes {
    int x3_dbg = env_int("KRK_X3_DBG");          // ablation bits of the split-bf16 conv / projection kernels (-DKRK_ABLATE builds)
    int lstm_dbg = env_int("KRK_LSTM_DBG");      // ablation bits of the recurrent kernels
    int lstm_v = env_int("KRK_LSTM_V", 0);       // 0: by hidden size (see recurrence_x3); 3: cluster kernel lstm_ws.hip; 1: streaming kernel
    int lstm_g = env_int("KRK_LSTM_G", 2);       // 4: four 16-line groups per cluster
    int lstm_m = env_int("KRK_LSTM_M");          // f32 plan: force 16- or 32-line tiles
    int conv_x6 = env_int("KRK_CONV_X6", 1);     // 0: the exact-f32 kernel also where the three-plane kernel (conv_x6.hip) is planned
    int taps_dma = env_int("KRK_TAPS_DMA", 1);   // conv_taps_x3.hip: input tile through raw-buffer -> LDS copies (0: register staging)
    int x3p_sb = env_int("KRK_X3P_SB", 1);       // 0: conv_x3p.hip with TWO tile buffers (80 instead of 52 KB of LDS: two workgroups per CU instead of three)
    int conv_x3p = env_int("KRK_CONV_X3P", 1);   // 0: conv_x3.hip also where the pipelined kernel (conv_x3p.hip) covers the geometry
};

constexpr int kFailed = -100;     // a step hit a hard error: the KRK_E_* code is in Pass::err, the message in g_err

// One forward call over a plan's schedule: what the steps share, and one method per step family.  The methods return a
// launcher code (0 ok, -4 unsupported configuration, other: launch failed) or kFailed.
struct Pass {
    krk_plan* p;
    int N;
    hipStream_t stream;
    const int* lens_host;
    bool one;                         // plain-bf16 plan: the _b1 launchers (cross terms compiled out)
    Probes probe;
    std::vector<int> Ws, Ns;          // tensor width and lines per length stage
    const int* d_lens = nullptr;      // [stage][N0] valid widths on the device (null: every line is full width)
    int N0 = 0;                       // lines of the call's input (N: lines of the running step's input)
    // stages behind a layer that changed the number of lines: the reference's seq_lens still count the INPUT's lines there
    // (Addition / Reshape hand them through), so the kernels see full-width lines ...
    std::vector<char> detached;
    std::vector<char> any_short;      // ... and this says whether the reference's GroupNorm would have met a short line
    bool front_done = false;
    int err = KRK_OK;

    const int* lens_at(int stage) const { return (d_lens && !detached[stage]) ? d_lens + (size_t)stage * N0 : nullptr; }
    int hard(int code, const std::string& msg) { err = fail(code, msg); return kFailed; }
    int nomem() { return hard(KRK_E_NOMEM, "forward: workspace allocation failed"); }
    int hip(hipError_t e, const char* what) {
        return e == hipSuccess ? 0 : hard(KRK_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    // marks the start of a profiled launch group (HIP event on the caller's stream)
    ~Pass() { end_groups(); }         // whatever path the call leaves by
    bool group_open = false;          // a roctx range of the running launch group is open (closed by the next mark / end_groups)
    void end_groups() {
        if (group_open) roctx().pop();
        group_open = false;
    }
    int mark(const char* name, double flops) {
        if (roctx().on()) {           // host-side span of the group's launches; the kernels carry the names rocprofv3 lists
            end_groups();
            roctx().push(name);
            group_open = true;
        }
        if (!p->profiling || p->prof_n + 1 >= p->events.size()) return 0;
        if (hipEventRecord(p->events[p->prof_n], stream) != hipSuccess) return hard(KRK_E_HIP, "hipEventRecord failed");
        p->prof_names[p->prof_n] = name;
        p->prof_flops[p->prof_n] = flops;
        ++p->prof_n;
        return 0;
    }

    int widths(int W);
    int upload_lens(int W);
    int conv(Step& s, const float* cur, float* outp, size_t out_elems, int Win);
    int layout(Step& s, const float* cur, float* outp, size_t out_elems, int Win, int Wout);
    int concat(Step& s, const std::vector<const float*>& outs, float* outp, int Wout);
    int split_input(Step& s, const float* cur, size_t in_elems, const void** xin);
    int projection(const ConvGeom& g, GemmX3Args& a);
    int linear(Step& s, const float* cur, float* outp, int Win);
    int lstm(Step& s, const float* cur, float* outp, size_t out_elems, int Win);
    int recurrence_x3(Step& s, float* outp, size_t out_elems, int N, int T, int G);
    int recurrence_f32(Step& s, float* outp, int N, int T, int G);
};

// per-stage tensor widths and line counts
int Pass::widths(int W) {
    N0 = N;
    if (const int bad = stage_dims(*p, N, W, Ns, Ws)) {
        const auto& op = p->lenops[bad - 1];
        if (op.kind == 6)
            return hard(KRK_E_INVALID, "forward: addition over the batch with chunk " + std::to_string(op.k) + " on " +
                                       std::to_string(Ns[op.from]) + " lines");
        return hard(KRK_E_INVALID, "forward: reshape does not divide a tensor of " + std::to_string(Ns[op.from]) + " lines of width " +
                                   std::to_string(Ws[op.from]));
    }
    detached.assign(p->nstages, 0);
    any_short.assign(p->nstages, 0);
    for (int s = 0; s + 1 < p->nstages; ++s) {
        if (Ws[s + 1] <= 0) return hard(KRK_E_INVALID, "forward: input width " + std::to_string(W) + " too small for this network");
        detached[s + 1] = detached[p->lenops[s].from] || Ns[s + 1] != Ns[p->lenops[s].from];
    }
    return 0;
}

// per-line valid widths of every stage, computed on the host, one asynchronous upload through a pinned buffer
int Pass::upload_lens(int W) {
    if (!lens_host) return 0;
    const size_t cnt = (size_t)p->nstages * N;
    if (cnt > p->h_lens_cap) {
        if (p->lens_ev_pending) { if (int r = hip(hipEventSynchronize(p->lens_ev), "hipEventSynchronize")) return r; p->lens_ev_pending = false; }
        if (p->h_lens_pinned) (void)hipHostFree(p->h_lens_pinned);
        p->h_lens_pinned = nullptr;
        if (int r = hip(hipHostMalloc((void**)&p->h_lens_pinned, cnt * sizeof(int) * 2, hipHostMallocDefault), "hipHostMalloc")) return r;
        p->h_lens_cap = cnt * 2;
    }
    if (p->d_lens.ensure(cnt * sizeof(int))) return hard(KRK_E_NOMEM, "forward: length table allocation failed");
    // the pinned staging buffer may still be in flight from the previous call
    if (p->lens_ev_pending) { if (int r = hip(hipEventSynchronize(p->lens_ev), "hipEventSynchronize")) return r; p->lens_ev_pending = false; }
    int* hl = p->h_lens_pinned;
    std::vector<int> raw;                  // a line's lengths per stage before the clamp to the tensor width
    for (int n = 0; n < N; ++n) {
        const int l = lens_host[n];
        if (l < 1 || l > W) return hard(KRK_E_INVALID, "forward: lens[" + std::to_string(n) + "] outside [1, W]");
        line_widths(*p, l, Ws, raw);
        hl[n] = l;
        if (l < W) any_short[0] = 1;
        for (int s = 0; s + 1 < p->nstages; ++s) {
            hl[(size_t)(s + 1) * N + n] = std::max(0, std::min(raw[s + 1], Ws[s + 1]));
            if (raw[s + 1] < Ws[s + 1]) any_short[s + 1] = 1;
        }
    }
    if (int r = hip(hipMemcpyAsync(p->d_lens.p, hl, cnt * sizeof(int), hipMemcpyHostToDevice, stream), "hipMemcpyAsync")) return r;
    if (int r = hip(hipEventRecord(p->lens_ev, stream), "hipEventRecord")) return r;
    p->lens_ev_pending = true;
    d_lens = (const int*)p->d_lens.p;
    return 0;
}

// ActConv2D (+ fused MaxPool / reshape): tap-as-K, split-bf16 implicit GEMM, split-bf16 first layer, or exact f32
int Pass::conv(Step& s, const float* cur, float* outp, size_t out_elems, int Win) {
    const ConvGeom& g = s.cg;
    if (g.taps) {
        ConvTapArgs a;
        a.pitch = nhcw_pitch(Win);
        a.x = (const __bf16*)cur; a.x_plane = (size_t)N * s.C * s.H * a.pitch;
        a.wpack = (const __bf16*)g.d_wx3; a.wpack5 = (const __bf16*)g.d_wx5; a.bias = g.d_b;
        a.y = (__bf16*)outp; a.y_plane = out_elems;
        a.len_out = lens_at(s.len_out);
        a.N = N; a.H = g.H; a.Cin = g.Cin; a.Cout = g.Cout; a.kh = g.kh; a.kw = g.kw; a.ph = g.ph; a.pw = g.pw;
        a.Ho = g.Ho; a.Wo = conv_out(Win, g.kw, g.sw, g.dw, g.pw);
        a.Hy = g.Hy; a.Wy = g.pool ? floordiv(a.Wo - 2, 2) + 1 : a.Wo;
        a.act = g.act;
        a.tiles_h = (g.Ho + 3) / 4; a.tiles_w = (a.Wo + 127) / 128;
        a.y_f32 = 0;
        a.dbg = probe.x3_dbg;
        a.dma = probe.taps_dma && (a.x_plane + (size_t)g.H * g.Cin * a.pitch) * 2 < 0xFFFFFF00ull ? 1 : 0;
        split_strides(g, a.Wo, a.Wy, a.y_sn, a.y_sr, a.y_sc);
        s.flops = 2.0 * N * (double)g.Ho * a.Wo * g.Cout * g.Cin * g.kh * g.kw;
        if (mark("conv_taps_x3", s.flops)) return kFailed;
        return one ? krk_launch_conv_taps_b1(a, g.pool, stream) : krk_launch_conv_taps(a, g.pool, stream);
    }
    if (g.x3) {
        X3Args a;
        fill_x3(g, a, cur, (size_t)N * s.C * s.H * Win, outp, N, Win, lens_at(s.len_in), lens_at(s.len_out), probe.x3_dbg);
        a.y_plane = out_elems;
        split_strides(g, a.Wo, a.Wy, a.y_sn, a.y_sr, a.y_sc);
        if (g.out_seq) {   // sequence rows for the projection: K-blocked
            a.y_cols = g.pool ? a.Wy : a.Wo;
            a.y_blkM = N * a.y_cols;
        }
        s.flops = 2.0 * N * (double)g.Ho * a.Wo * g.Cout * g.Cin * g.kh * g.kw;
        if (mark("conv_x3", s.flops)) return kFailed;
        if (g.xtps > 0 && probe.conv_x3p && (size_t)g.H * Win * g.Cin * 2 < 0x7fffffffull) {
            a.tps = g.xtps;
            a.single_buf = probe.x3p_sb;
            return one ? krk_launch_conv_x3p_b1(a, g.pool, stream) : krk_launch_conv_x3p(a, g.pool, stream);
        }
        return one ? krk_launch_conv_x3_b1(a, false, g.pool, stream) : krk_launch_conv_x3(a, false, g.pool, stream);
    }
    if (g.c1x3) {
        Conv1Args a;
        a.x = cur; a.wpack = (const __bf16*)g.d_wx3; a.bias = g.d_b;
        a.y = (__bf16*)outp; a.y_plane = out_elems;
        a.len_in = lens_at(s.len_in); a.len_out = lens_at(s.len_out);
        a.N = N; a.H = g.H; a.W = Win; a.Cout = g.Cout; a.kh = g.kh; a.kw = g.kw; a.ph = g.ph; a.pw = g.pw;
        a.Cin = g.Cin;
        a.Ho = g.Ho; a.Wo = conv_out(Win, g.kw, g.sw, g.dw, g.pw);
        a.Hy = g.Hy; a.Wy = g.pool ? floordiv(a.Wo - 2, 2) + 1 : a.Wo;
        a.act = g.act;
        a.tiles_h = (g.Ho + 7) / 8; a.tiles_w = (a.Wo + 127) / 128;
        split_strides(g, a.Wo, a.Wy, a.y_sn, a.y_sr, a.y_sc);
        a.y_pitch = g.out_nhcw ? nhcw_pitch(a.Wy) : 0;
        a.y_f32 = 0;
        a.dbg = probe.x3_dbg;
        s.flops = 2.0 * N * (double)g.Ho * a.Wo * g.Cout * g.Cin * g.kh * g.kw;
        if (mark("conv1_x3", s.flops)) return kFailed;
        for (int hf = 0; 32 * hf < g.Cout; ++hf) {             // 32 filters per launch: the halves of the channels-last output
            Conv1Args h = a;
            h.wpack = a.wpack + (size_t)hf * g.Cin * g.kh * 2 * 64 * 8;
            h.bias = a.bias + 32 * hf;
            h.y = a.y + 32 * hf;
            h.Cout = std::min(32, g.Cout - 32 * hf);
            if (int rc = one ? krk_launch_conv1_x3_b1(h, g.pool, stream) : krk_launch_conv1_x3(h, g.pool, stream)) return rc;
        }
        return 0;
    }
    if (g.x6 && probe.conv_x6) {
        // fp32 NCHW -> three bf16 planes NHWC (h + m + l = x to 2^-24), then the six-term convolution; fp32 NCHW out
        const size_t in_elems = (size_t)N * s.C * s.H * Win;
        if (s.aux2.ensure(in_elems * 3 * sizeof(uint16_t))) return nomem();
        if (mark("split3", 0)) return kFailed;
        if (int rc = krk_launch_split3_nhwc(cur, s.aux2.p, in_elems, N, s.C, s.H, Win, stream)) return rc;
        X3Args a;
        fill_x3(g, a, s.aux2.p, in_elems, outp, N, Win, lens_at(s.len_in), lens_at(s.len_out), probe.x3_dbg);
        a.wpack = (const __bf16*)g.d_wx6;
        a.CBpad = g.x6CBpad;
        s.flops = 2.0 * N * (double)g.Ho * a.Wo * g.Cout * g.Cin * g.kh * g.kw;
        if (mark("conv_x6", s.flops)) return kFailed;
        return krk_launch_conv_x6(a, g.pool, stream);
    }
    ConvArgs a;
    fill_conv(g, a, cur, outp, N, Win, lens_at(s.len_in), lens_at(s.len_out));
    if (g.split_out) {
        a.y_split = outp;
        a.y_plane = out_elems;
        split_strides(g, a.Wo, a.Wy, a.y_sn, a.y_sr, a.y_sc);
        if (g.out_nhcw) {      // [N][H][C][pitch] planes for the tap kernel, which reads whole pitched rows: the columns between the
            const long pitch = nhcw_pitch(a.Wy);       // tensor width and the pitch are zeroed here (the kernel stores columns < Wy only)
            a.y_sn = (long)g.Hy * g.Cout * pitch; a.y_sr = (long)g.Cout * pitch; a.y_sc = 1; a.y_cs = pitch;
            if (pitch > a.Wy)
                if (int r = hip(hipMemset2DAsync((__bf16*)outp + a.Wy, (size_t)pitch * 2, 0, (size_t)(pitch - a.Wy) * 2,
                                                 2 * (size_t)N * g.Hy * g.Cout, stream), "hipMemset2DAsync")) return r;
        }
    }
    s.flops = 2.0 * N * (double)g.Ho * a.Wo * g.Cout * g.Cin * g.kh * g.kw;
    if (mark("conv", s.flops)) return kFailed;
    return krk_launch_conv(a, false, g.out_seq, g.pool, stream);
}

// the steps without arithmetic of their own: MaxPool, GroupNorm, reshapes and layout changes
int Pass::layout(Step& s, const float* cur, float* outp, size_t out_elems, int Win, int Wout) {
    s.flops = 0;
    switch (s.kind) {
        case S_MAXPOOL:
            if (s.on_split) {
                if (mark("maxpool_x3", 0)) return kFailed;
                return krk_launch_maxpool_x3(cur, (size_t)N * s.C * s.H * Win, outp, out_elems, lens_at(s.len_out), N, s.C, s.H, Win,
                                             s.kh, s.kw, s.sh, s.sw, s.Ho, Wout, stream);
            }
            if (mark("maxpool", 0)) return kFailed;
            return krk_launch_maxpool(cur, outp, lens_at(s.len_out), N, s.C, s.H, Win, s.kh, s.kw, s.sh, s.sw, s.Ho, Wout, stream);
        case S_GN: {
            const bool pool = s.pooled;
            if (s.c1gn) {     // `cur` is the INPUT IMAGE: the convolution in front is recomputed inside both passes
                if (mark(pool ? "conv1_groupnorm_pool" : "conv1_groupnorm", 2.0 * N * (double)s.H * Win * s.C * 9)) return kFailed;
                C1GnArgs a;
                a.x = cur; a.w = s.d_c1w; a.bias = s.d_c1b; a.gamma = s.d_gamma; a.beta = s.d_beta;
                a.lens = lens_at(s.len_in); a.len_out = pool ? lens_at(s.len_out) : nullptr;
                a.N = N; a.C = s.C; a.H = s.H; a.W = Win; a.G = s.groups; a.act = s.c1_act; a.pool = pool ? 1 : 0;
                a.Ho = pool ? s.Ho : s.H; a.Wo = pool ? Wout : Win;
                a.eps = 1e-5f;
                a.chunks = krk_c1gn_chunks(N, s.H);
                if (s.aux.ensure((size_t)2 * N * s.groups * a.chunks * sizeof(double))) return nomem();
                a.part = (double*)s.aux.p;
                a.y = outp;
                return krk_launch_c1gn(a, stream);
            }
            if (mark(pool ? "groupnorm_pool" : "groupnorm", 0)) return kFailed;
            const int chunks = krk_groupnorm_chunks(N, s.C, s.H, Win, s.groups, pool ? s.Ho : 0);
            if (s.aux.ensure((size_t)2 * N * s.groups * chunks * sizeof(double))) return nomem();
            return krk_launch_groupnorm(cur, outp, s.d_gamma, s.d_beta, lens_at(s.len_in), pool ? lens_at(s.len_out) : nullptr, N, s.C,
                                        s.H, Win, s.groups, 1e-5f, pool ? s.kh : 0, s.kw, s.sh, s.sw, s.Ho, Wout, (double*)s.aux.p,
                                        stream);
        }
        case S_TOSEQ:
            if (s.on_split) {
                if (mark("to_seq_x3", 0)) return kFailed;
                return krk_launch_toseq_x3(cur, outp, out_elems, N, s.C, s.H, Win, stream);
            }
            if (s.split_rows) {
                if (mark("to_seq_split", 0)) return kFailed;
                return krk_launch_toseq_split_f32(cur, outp, out_elems, N, s.C, s.H, Win, stream);
            }
            if (mark("to_seq", 0)) return kFailed;
            return krk_launch_to_seq(cur, outp, N, s.C, s.H, Win, stream);
        case S_UNSPLIT:
            if (mark("unsplit", 0)) return kFailed;
            return krk_launch_unsplit(cur, out_elems, outp, N, s.C, s.H, Win, stream);
        case S_IMG2ROWS:
            if (mark("img2rows", 0)) return kFailed;
            return krk_launch_img2rows(cur, outp, N, s.C, s.H, Win, s.yaxis, stream);
        case S_ROWS2IMG:
            if (mark("rows2img", 0)) return kFailed;
            return krk_launch_rows2img(cur, outp, N, s.C, s.H, Win, s.yaxis, s.yaxis ? lens_at(s.len_in) : nullptr, s.last_only, stream);
        case S_UPZERO:
            if (mark("zero_insert", 0)) return kFailed;
            return krk_launch_upzero(cur, outp, (size_t)N * s.C, s.H, Win, s.sh, s.sw, s.outH, Wout, stream);
        case S_SOFTMAXC:
            if (mark("softmax", 0)) return kFailed;
            return krk_launch_softmax_c(cur, outp, N, s.C, s.H, Win, lens_at(s.len_in), stream);
        case S_ADD: {
            if (mark("add", 0)) return kFailed;
            // channels of an image: N blocks of C*H*W, pieces of chunk*H*W; channels of sequence rows: N*T rows of C, pieces of
            // chunk; height: N*C planes of H*W, pieces of chunk*W
            if (s.add_axis == 3) {      // lines: one block of N * (C H W), pieces of chunk lines (image or sequence rows alike)
                const size_t line = (size_t)s.C * s.H * Win;
                return krk_launch_chunk_sum(cur, outp, 1, (size_t)s.chunk * line, N / s.chunk, (size_t)N * line, stream);
            }
            if (s.add_axis == 2) {
                if (Win < s.chunk) return hard(KRK_E_INVALID, "forward: addition over the width with chunk " + std::to_string(s.chunk) +
                                                              " on a tensor of " + std::to_string(Win) + " columns");
                return krk_launch_chunk_sum(cur, outp, (size_t)N * s.C * s.H, (size_t)s.chunk, Win / s.chunk, (size_t)Win, stream);
            }
            if (s.add_axis == 1)
                return krk_launch_chunk_sum(cur, outp, (size_t)N * s.C, (size_t)s.chunk * Win, s.nk, (size_t)s.H * Win, stream);
            if (s.out_is_seq)
                return krk_launch_chunk_sum(cur, outp, (size_t)N * Win, (size_t)s.chunk, s.nk, (size_t)s.C, stream);
            return krk_launch_chunk_sum(cur, outp, (size_t)N, (size_t)s.chunk * s.H * Win, s.nk, (size_t)s.C * s.H * Win, stream);
        }
        case S_PERMUTE: {
            if (mark("reshape", 0)) return kFailed;
            const krk_plan::ReshapeOp& r = p->reshapes[s.reshape];
            const int in[4] = {N, s.C, s.H, Win};
            int d5[5], perm[5], out[4];
            if (!reshape_dims(r, in, d5, perm, out)) return hard(KRK_E_INVALID, "forward: reshape does not divide the tensor");
            if (out[1] != s.outC || out[2] != s.outH)
                return hard(KRK_E_INVALID, "forward: reshape gives " + std::to_string(out[1]) + " channels x " + std::to_string(out[2]) +
                                           " rows for this batch, the layers behind it were built for " + std::to_string(s.outC) + " x " +
                                           std::to_string(s.outH) + " (the reference derives them from the spec's input shape)");
            // element strides of the input's four axes: fp32 NCHW, or the rows (N, T, C) of a sequence layer
            const size_t st4[4] = {(size_t)s.C * s.H * Win, s.in_seq ? (size_t)1 : (size_t)s.H * Win, (size_t)Win,
                                   s.in_seq ? (size_t)s.C : (size_t)1};
            size_t st5[5];
            for (int i = 0, j = 0; i < 4; ++i) {
                if (i == r.src) { st5[j] = st4[i] * d5[j + 1]; st5[j + 1] = st4[i]; j += 2; }
                else st5[j++] = st4[i];
            }
            int pd[5];
            size_t ps[5];
            for (int i = 0; i < 5; ++i) { pd[i] = d5[perm[i]]; ps[i] = st5[perm[i]]; }
            return krk_launch_permute5(cur, outp, pd, ps, stream);
        }
        default:
            return -4;
    }
}

// MultiParamParallel: torch.cat(outputs, dim=1) of the members' fp32 outputs (reference layers.py:70)
int Pass::concat(Step& s, const std::vector<const float*>& outs, float* outp, int Wout) {
    if (mark("concat", 0)) return kFailed;
    size_t coff = 0;
    for (const auto& m : s.srcs) {
        if (Ws[m.stage] != Wout)       // torch.cat would refuse: the members' widths differ for this input width
            return hard(KRK_E_INVALID, "forward: parallel group members produce different widths (" + std::to_string(Ws[m.stage]) +
                                           " and " + std::to_string(Wout) + ")");
        if (Ns[m.stage] != N)
            return hard(KRK_E_INVALID, "forward: parallel group members produce different numbers of lines (" + std::to_string(Ns[m.stage]) +
                                           " and " + std::to_string(N) + ")");
        const float* src = m.step < 0 ? nullptr : outs[m.step];
        if (!src) return hard(KRK_E_INVALID, "forward: parallel group member without an output");
        // image: N blocks of C_k*H*W floats into blocks of C*H*W; sequence rows (N, T, C): N*T blocks of C_k into rows of C
        const size_t unit = s.out_is_seq ? 1 : (size_t)s.H * Wout;
        const size_t outer = s.out_is_seq ? (size_t)N * Wout : (size_t)N;
        if (int rc = krk_launch_concat(src, outp, outer, m.C * unit, s.C * unit, coff * unit, stream)) return rc;
        coff += m.C;
    }
    return 0;
}

// fp32 rows (from an f32 producer) -> the split planes the bf16x3 projection reads; a no-op when they arrive split
int Pass::split_input(Step& s, const float* cur, size_t in_elems, const void** xin) {
    *xin = cur;
    if (s.in_split) return 0;
    if (s.aux2.ensure(in_elems * sizeof(float))) return nomem();
    if (mark("split", 0)) return kFailed;
    const int kx = s.cg.xK ? s.cg.xK : s.cg.Cin, rows = (int)(in_elems / kx);
    if (kx != s.cg.Cin) {         // the pad octets of both planes (the kernel writes ceil(Cin / 8) octets of kx / 8, the last one zero-filled)
        const int done = (s.cg.Cin + 7) / 8, pad = kx / 8 - done;
        if (pad > 0)
            if (int r = hip(hipMemset2DAsync((char*)s.aux2.p + (size_t)done * rows * 16, in_elems * 2, 0, (size_t)pad * rows * 16, 2, stream), "hipMemset2DAsync")) return r;
    }
    if (int rc = krk_launch_split_rows(cur, s.aux2.p, rows, s.cg.Cin, in_elems, stream)) return rc;
    *xin = s.aux2.p;
    return 0;
}

// split-bf16 row projection (gemm_x3.hip).  Round 4's wide-tile variant (gemm_x3w.hip: 256 x 320 tiles, eight waves) won 9 % alone and
// lost 1.7 % on the pipelined bench -- one 110 KB workgroup per CU shuts the other batches' kernels out -- and left the tree in round 5
// (git show be5f925:kraken_amd/csrc/gemm_x3w.hip).
int Pass::projection(const ConvGeom& g, GemmX3Args& a) {
    return one ? krk_launch_gemm_x3_b1(a, stream) : krk_launch_gemm_x3(a, stream);
}

// LinSoftmax's projection (logits; the softmax belongs to the decode)
int Pass::linear(Step& s, const float* cur, float* outp, int Win) {
    s.flops = 2.0 * N * (double)Win * s.cg.Cout * s.cg.Cin;
    if (s.cg.x3) {
        const int Nr = s.in_tiled ? (N + 15) / 16 * 16 : N;      // tile-time-major rows come in whole 16-line tiles
        const size_t in_elems = (size_t)Nr * Win * (s.cg.xK ? s.cg.xK : s.cg.Cin);      // elements per plane, the pad octet included
        const void* xin;
        if (int rc = split_input(s, cur, in_elems, &xin)) return rc;
        GemmX3Args a;
        fill_gemm(s.cg, a, xin, in_elems, outp, Nr * Win, probe.x3_dbg);
        if (s.in_tiled) { a.tileT = -Win; a.nlines = N; }          // back to line-major rows for the decode
        if (mark("linear_x3", s.flops)) return kFailed;
        return projection(s.cg, a);
    }
    ConvArgs a;
    fill_conv(s.cg, a, cur, outp, 1, N * Win, nullptr, nullptr);
    if (mark("linear", s.flops)) return kFailed;
    return krk_launch_conv(a, true, true, false, stream);
}

// TransposedSummarizingRNN: the input projection of every step as one GEMM, then the recurrence
int Pass::lstm(Step& s, const float* cur, float* outp, size_t out_elems, int Win) {
    if (s.img_axis == 1 && lens_host)
        return hard(KRK_E_UNSUPPORTED, "forward: seq_lens with an LSTM over image rows (the reference raises too, layers.py:528-530)");
    // sequences and steps: plain (N, W); image rows (N*H, W); image columns (N*W, H)
    const int Himg = s.outH;
    const int T = s.img_axis == 2 ? Himg : Win;
    const int Ns = s.img_axis == 1 ? N * Himg : (s.img_axis == 2 ? N * Win : N);
    const int G = 4 * s.Hp;
    // tile-time-major projections (bf16x3 recurrence) address whole 16-line tiles: pad the row count
    const size_t xp_elems = (size_t)(s.rec_x3 ? (Ns + 15) / 16 * 16 : Ns) * T * s.ndir * G;
    if (s.aux.ensure(xp_elems * sizeof(float))) return nomem();
    const double xflops = 2.0 * Ns * (double)T * s.ndir * 4.0 * s.hidden * s.cg.Cin;
    int rc;
    if (s.cg.x3) {
        const int Nr = s.in_tiled ? (Ns + 15) / 16 * 16 : Ns;
        const size_t in_elems = (size_t)Nr * T * (s.cg.xK ? s.cg.xK : s.cg.Cin);
        const void* xin;
        if ((rc = split_input(s, cur, in_elems, &xin))) return rc;
        GemmX3Args a;
        fill_gemm(s.cg, a, xin, in_elems, (float*)s.aux.p, Nr * T, probe.x3_dbg);
        // projection rows for the split-bf16 recurrence are tile-time-major: permute line-major input rows, keep tiled ones;
        // an f32 recurrence behind tiled input gets its line-major rows back
        a.tileT = s.rec_x3 ? (s.in_tiled ? 0 : T) : (s.in_tiled ? -T : 0);
        a.nlines = Ns;
        if (mark("lstm_xproj_x3", xflops)) return kFailed;
        rc = projection(s.cg, a);
    } else {
        ConvArgs a;
        fill_conv(s.cg, a, cur, (float*)s.aux.p, 1, Ns * T, nullptr, nullptr);
        if (mark("lstm_xproj", xflops)) return kFailed;
        rc = krk_launch_conv(a, true, true, false, stream);
    }
    if (rc) return rc;
    if (!front_done) {   // convolution block + first projection are enqueued: the next batch may start its own
        if ((rc = hip(hipEventRecord(p->front_ev, stream), "hipEventRecord"))) return rc;
        front_done = true;
    }
    if (mark(s.rec_x3 ? "lstm_rec_x3" : "lstm_rec", 2.0 * Ns * (double)T * s.ndir * 4.0 * s.hidden * s.hidden)) return kFailed;
    if (lens_host && (rc = hip(hipMemsetAsync(outp, 0, out_elems * sizeof(float), stream), "hipMemsetAsync"))) return rc;
    s.flops = 2.0 * Ns * (double)T * s.ndir * 4.0 * s.hidden * ((double)s.cg.Cin + s.hidden);
    return s.rec_x3 ? recurrence_x3(s, outp, out_elems, Ns, T, G) : recurrence_f32(s, outp, Ns, T, G);
}

// split-bf16 recurrence: the weight-stationary cluster kernel, or the streaming kernel for shapes outside its range
int Pass::recurrence_x3(Step& s, float* outp, size_t out_elems, int Ns, int T, int G) {
    LstmX3Args l;
    l.xp = (const float*)s.aux.p;
    l.wp = (const __bf16*)s.d_wrecx3;
    l.out = (__bf16*)outp;
    l.out_plane = out_elems;
    l.lens = lens_at(s.len_in);
    l.N = Ns; l.T = T; l.H = s.opad ? s.Hp : s.hidden; l.Hp = s.Hp; l.G = G;     // (H only shapes the stores: Step::opad)
    l.NB = G / 16; l.NKB = (s.Hp + 31) / 32;
    l.ndir = s.ndir; l.dirmode = s.dirmode;
    l.xstride = s.ndir * G;
    l.ostride = s.ndir * (s.opad ? s.Hp : s.hidden);
    l.hrow = l.NKB * 64 + 16;
    l.xtiled = 1;
    l.otiled = s.out_tiled ? 1 : 0;
    l.dbg = probe.lstm_dbg;
    // Which kernel (tools/lstm_narrow_probe.py, profiles/r04_lstm_narrow_probe.txt; N = 256, T = 150, ms per launch):
    //   H     16     32     64     96    128    200
    //   x3   0.15   0.13   0.22   0.36   0.51   1.29     streaming kernel (lstm_x3.hip): W_hh through L1 every step
    //   ws   0.31   0.30   0.29   0.31   0.32   0.46     weight-stationary 4-CU clusters (lstm_ws.hip)
    // Up to 64 hidden units the recurrent weights (<= 64 KB split) stream faster than a cluster exchanges; above, the cluster kernel.
    // (Round 3's third kernel, lstm_wp.hip -- XCD-local clusters, deferred gates, gather waves -- tied lstm_ws at H = 200 and lost
    // to one of the two everywhere else: 0.22 / 0.24 / 0.26 / 0.33 / 0.36 ms; removed in round 4)
    // the plan's own setting (krk_plan_set_recurrence: the one retry after an exchange timeout) wins over the process-wide probe
    const int lstm_v = p->recurrence == KRK_RECURRENCE_STREAMING ? 1 : probe.lstm_v ? probe.lstm_v : (s.Hp <= 64 ? 1 : 3);
    if (!s.d_wrecws || lstm_v != 3) return krk_launch_lstm_x3(l, stream);
    // 16-line groups per cluster: 2.  With 4 (64 lines on 4 CUs, the exchange three slots old when read) the slot time barely
    // moves (it is not exchange bound), so a launch takes twice as long on half the CUs: same chip time, worse latency, fewer
    // lines/s through the pipelined engine.  KRK_LSTM_G=4 keeps it probeable.
    const int groups = probe.lstm_g == 4 ? 4 : 2;
    const size_t gbytes = std::max(krk_lstm_ws_gran_bytes(Ns, s.ndir, s.ws_bpc, 4), krk_lstm_ws_gran_bytes(Ns, s.ndir, s.ws_bpc, 2));
    if (gbytes > s.ws_gran.cap) {
        if (s.ws_gran.ensure(gbytes)) return nomem();
        if (int r = hip(hipMemsetAsync(s.ws_gran.p, 0, s.ws_gran.cap, stream), "hipMemsetAsync")) return r;   // tags of a fresh buffer must not match
    }
    LstmWsArgs w;
    w.xp = l.xp; w.wp = (const __bf16*)s.d_wrecws; w.out = l.out; w.out_plane = l.out_plane; w.lens = l.lens;
    w.N = l.N; w.T = l.T; w.H = l.H; w.Hp = l.Hp; w.NKB = l.NKB; w.NB = l.NB; w.G = l.G;
    w.ndir = l.ndir; w.dirmode = l.dirmode; w.xstride = l.xstride; w.ostride = l.ostride; w.hrow = l.hrow;
    w.BPC = s.ws_bpc;
    w.gran = (unsigned long long*)s.ws_gran.p;
    // the ticket counter is zeroed before every launch (a 64-byte memset node): a monotonic counter mirrored on the host is one
    // missed bump away from shifting every cluster of every later launch
    if (int r = hip(hipMemsetAsync(s.ws_ctrl, 0, 64, stream), "hipMemsetAsync")) return r;
    w.ctrl = s.ws_ctrl;
    w.ticket_base = 0;
    s.ws_epoch = s.ws_epoch % 65535u + 1u;
    w.epoch = s.ws_epoch;
    w.err = p->err_dev;
    w.otiled = l.otiled;
    w.dbg = l.dbg;
    int rc = one ? krk_launch_lstm_ws_b1(w, groups, stream) : krk_launch_lstm_ws(w, groups, stream);
    if (rc == -4) rc = krk_launch_lstm_x3(l, stream);
    return rc;
}

// exact-f32 recurrence: register-resident kernel for tiny hidden sizes, else 16- or 32-line MFMA tiles
int Pass::recurrence_f32(Step& s, float* outp, int Ns, int T, int G) {
    LstmArgs l;
    l.xp = (const float*)s.aux.p;
    l.out = outp;
    l.lens = s.img_axis ? nullptr : lens_at(s.len_in);   // image columns/rows always run their full length
    l.N = Ns; l.T = T; l.H = s.hidden; l.Hp = s.Hp; l.G = G;
    l.ndir = s.ndir; l.dirmode = s.dirmode;
    l.xstride = s.ndir * G;
    l.ostride = s.ndir * s.hidden;
    l.dbg = probe.lstm_dbg;
    l.peep = s.d_peep;
    if (s.d_wrecsm) {   // hidden size <= 32: one wave per 16 sequences, weights / h / c in registers
        l.wp = s.d_wrecsm;
        l.NG = l.NB = 0;
        const int rc = s.d_wrecsmx ? krk_launch_lstm_small_x3(l, s.d_wrecsmx, stream) : krk_launch_lstm_small(l, stream);
        if (rc != -4) return rc;            // (an output of 2 GiB or more: the generic kernel below)
    }
    if (s.Hp > 256 || s.d_peep) {   // generic-width kernel: 16-line tiles, K groups of 4 steps (krk_lstm_kg(16, > 13 blocks per wave) == 4)
        l.wp = s.d_wrec16;
        l.NB = G / 16;
        l.NG = (s.Hp / 4 + 3) / 4;
        // above the widths LDS holds (768 / 1152): the workgroups' cell state, then h too, in this step's scratch (lstm_rec.hip)
        const size_t cfl = krk_lstm_big_cstate_floats(Ns, s.ndir, s.Hp), hfl = krk_lstm_big_hstate_floats(Ns, s.ndir, s.Hp);
        if (cfl + hfl) {
            if (s.ws_gran.ensure((cfl + hfl) * sizeof(float))) return nomem();      // (the cluster kernel's buffer: never both on one step)
            l.cstate = (float*)s.ws_gran.p;
            if (hfl) {
                l.hstate = (float*)s.ws_gran.p + cfl;
                if (int r = hip(hipMemsetAsync(l.hstate, 0, hfl * sizeof(float), stream), "hipMemsetAsync")) return r;
            }
        }
        return krk_launch_lstm_big(l, stream);
    }
    // 32-line tiles once they fill most of the 256 CUs, 16-line tiles below that
    const int tiles32 = (Ns + 31) / 32 * s.ndir;
    const int M_auto = tiles32 >= 192 ? 32 : 16;
    const int M = (probe.lstm_m == 16 || probe.lstm_m == 32) ? probe.lstm_m : M_auto;
    l.wp = (M == 32) ? s.d_wrec32 : s.d_wrec16;
    const int ks = s.Hp / ((M == 32) ? 2 : 4);
    const int kg = krk_lstm_kg(M, (G / M + 3) / 4);
    l.NG = (ks + kg - 1) / kg;
    l.NB = G / M;
    return krk_launch_lstm(l, M, stream);
}

// Runs the schedule.  `final_out` (may be null) receives the last step's output.
// On return *final_ptr points at the last step's output buffer and *d_olens at the
// device copy of the final valid widths (or null when lens_host is null).
int run_plan(krk_plan* p, const float* x_dev, const int* lens_host, int N, int W, hipStream_t stream,
             float* final_out, const float** final_ptr, const int** d_olens, int* T_out) {
    if (!p || !x_dev) return fail(KRK_E_INVALID, "forward: null plan or input");
    if (N <= 0 || W <= 0) return fail(KRK_E_INVALID, "forward: N and W must be positive");
    HIPCHK(hipSetDevice(p->device));
    p->last_N = N;
    p->last_W = W;
    if (p->front_wait) {   // one-shot: this batch's convolution block starts after the other plan's has finished
        HIPCHK(hipStreamWaitEvent(stream, p->front_wait, 0));
        p->front_wait = nullptr;
    }
    RoctxRange whole(roctx().on() ? "krk_forward N=" + std::to_string(N) + " W=" + std::to_string(W) : std::string());
    Pass pass{p, N, stream, lens_host, p->precision == KRK_PREC_BF16};
    if (pass.widths(W) || pass.upload_lens(W)) return pass.err;

    const float* cur = x_dev;
    const size_t nsteps = p->steps.size();
    std::vector<const float*> outs(nsteps, nullptr);      // where every step's output lives (parallel groups read them again)
    p->prof_n = 0;
    if (p->profiling) {
        p->prof_names.assign(p->events.size(), nullptr);
        p->prof_flops.assign(p->events.size(), 0.0);
    }
    for (size_t si = 0; si < nsteps; ++si) {
        Step& s = p->steps[si];
        const int Win = pass.Ws[s.len_in], Wout = pass.Ws[s.len_out];
        const bool is_last = (si + 1 == nsteps);
        pass.N = pass.Ns[s.len_in];                 // lines of this step's input (an Addition / Reshape on the batch axis changes them)
        const int Nout = pass.Ns[s.len_out];
        size_t out_elems = (size_t)Nout * s.outC * s.outH * Wout;
        if (s.kind == S_CONV && s.cg.out_nhcw) out_elems = (size_t)Nout * s.outC * s.outH * nhcw_pitch(Wout);
        if (s.kind == S_LSTM && s.out_tiled) out_elems = (size_t)((Nout + 15) / 16 * 16) * s.outC * s.outH * Wout;
        const size_t seq_feat = (size_t)s.outC * s.outH;             // features of a sequence row
        if (s.seq_kpad) out_elems = out_elems / seq_feat * s.seq_kpad;   // planes of seq_kpad features: one zeroed octet behind the real ones
        // behind a layer that changed the number of lines the reference's seq_lens no longer match the tensor: its packed LSTM
        // (pack_padded_sequence) and its masked GroupNorm (layers.py:977-984, unless every line is full width) raise
        if (lens_host && pass.detached[s.len_in]) {
            if (s.kind == S_LSTM && !s.img_axis)
                return fail(KRK_E_INVALID, "forward: seq_lens of " + std::to_string(pass.N0) + " lines reach a recurrent layer behind a "
                                           "batch-changing layer (" + std::to_string(pass.N) + " lines): the reference raises here too");
            if (s.kind == S_GN && pass.any_short[s.len_in])
                return fail(KRK_E_INVALID, "forward: seq_lens of " + std::to_string(pass.N0) + " lines with padding reach a GroupNorm behind "
                                           "a batch-changing layer (" + std::to_string(pass.N) + " lines): the reference raises here too");
        }
        if (s.kind == S_ALIAS) {      // a parallel member starts from the tensor in front of its group
            cur = s.src < 0 ? x_dev : outs[s.src];
            outs[si] = cur;
            continue;
        }
        float* outp;
        if (is_last && final_out) outp = final_out;
        else {
            if (s.out.ensure(out_elems * sizeof(float))) return fail(KRK_E_NOMEM, "forward: workspace allocation failed");
            outp = (float*)s.out.p;
        }
        if (s.seq_kpad && !s.skip) {
            const size_t rows = out_elems / s.seq_kpad;
            // the octets behind the last full one: a partly real octet (features not a multiple of 8: the recurrence writes its real
            // elements afterwards) and the pad octet
            HIPCHK(hipMemset2DAsync((char*)outp + (seq_feat / 8) * rows * 16, out_elems * 2, 0, (s.seq_kpad / 8 - seq_feat / 8) * rows * 16, 2, stream));
        }
        if (s.skip) { outs[si] = cur; continue; }   // recomputed inside the next step (c1gn.hip): `cur` stays the step's input
        int rc;
        switch (s.kind) {
            case S_CONV: rc = pass.conv(s, cur, outp, out_elems, Win); break;
            case S_LINEAR: rc = pass.linear(s, cur, outp, Win); break;
            case S_LSTM: rc = pass.lstm(s, cur, outp, out_elems, Win); break;
            case S_CONCAT: rc = pass.concat(s, outs, outp, Wout); break;
            default: rc = pass.layout(s, cur, outp, out_elems, Win, Wout); break;
        }
        if (rc == kFailed) return pass.err;
        if (rc == -4) return fail(KRK_E_UNSUPPORTED, std::string("forward: unsupported kernel configuration in ") + kStepNames[s.kind]);
        if (rc) return fail(KRK_E_HIP, std::string("forward: launch of ") + kStepNames[s.kind] + " failed: " +
                                           hipGetErrorString(hipGetLastError()));
        cur = outp;
        outs[si] = outp;
    }
    if (!pass.front_done) HIPCHK(hipEventRecord(p->front_ev, stream));
    if (p->profiling) HIPCHK(hipEventRecord(p->events[p->prof_n], stream));
    if (final_ptr) *final_ptr = cur;
    if (d_olens) *d_olens = pass.lens_at(p->out_stage);
    if (T_out) *T_out = pass.Ws[p->out_stage];
    return KRK_OK;
}

int decode_on_device(const float* scores, long sn, long sc, long st, int N, int C, int T, const int* d_olens,
                     int softmax, float temperature, float* probs, int* d_labels, float* d_confs,
                     hipStream_t stream, const krk_decode_out* out) {
    if (!out || !out->labels || !out->starts || !out->ends || !out->confs || !out->counts)
        return fail(KRK_E_INVALID, "decode: null output buffers");
    if (out->t_stride < T) return fail(KRK_E_INVALID, "decode: t_stride < T");
    if (softmax && !(temperature > 0.f)) return fail(KRK_E_INVALID, "decode: temperature must be > 0");
    if ((size_t)T * 2 * sizeof(float) > 60 * 1024) return fail(KRK_E_UNSUPPORTED, "decode: more than 7680 time steps");
    int rc = krk_launch_rowmax(scores, sn, sc, st, N, C, T, softmax, temperature, probs, d_labels, d_confs, stream);
    if (!rc)
        rc = krk_launch_collapse(d_labels, d_confs, d_olens, N, T, out->labels, out->starts, out->ends, out->confs,
                                 out->counts, out->t_stride, stream);
    if (rc) return fail(KRK_E_HIP, std::string("decode: kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return KRK_OK;
}

}  // namespace

extern "C" {

int krk_forward(krk_plan* plan, const float* x_dev, const int* lens_host, int N, int W, void* stream,
                float* out_dev) {
    if (!out_dev) return fail(KRK_E_INVALID, "krk_forward: out_dev is null");
    return run_plan(plan, x_dev, lens_host, N, W, (hipStream_t)stream, out_dev, nullptr, nullptr, nullptr);
}

int krk_greedy_decode(const float* scores_dev, long sn, long sc, long st, int N, int C, int T,
                      const int* olens_host, int softmax, float temperature, float* probs_dev, void* stream,
                      const krk_decode_out* out) {
    if (!scores_dev || N <= 0 || C <= 0 || T <= 0) return fail(KRK_E_INVALID, "krk_greedy_decode: bad argument");
    if (krk_device_count() <= 0) return fail(KRK_E_HIP, "krk_greedy_decode: no HIP device");
    hipStream_t s = (hipStream_t)stream;
    if (olens_host)
        for (int n = 0; n < N; ++n)
            if (olens_host[n] < 0 || olens_host[n] > T) return fail(KRK_E_INVALID, "krk_greedy_decode: olens outside [0, T]");
    // stream-ordered scratch, released on every path
    struct Scratch {
        hipStream_t s;
        void* p[3] = {nullptr, nullptr, nullptr};
        ~Scratch() { for (void* q : p) if (q) (void)hipFreeAsync(q, s); }
    } tmp{s};
    const size_t rows = (size_t)N * T;
    HIPCHK(hipMallocAsync(&tmp.p[0], rows * sizeof(int), s));
    HIPCHK(hipMallocAsync(&tmp.p[1], rows * sizeof(float), s));
    if (olens_host) {
        HIPCHK(hipMallocAsync(&tmp.p[2], (size_t)N * sizeof(int), s));
        HIPCHK(hipMemcpyAsync(tmp.p[2], olens_host, (size_t)N * sizeof(int), hipMemcpyHostToDevice, s));
    }
    return decode_on_device(scores_dev, sn, sc, st, N, C, T, (const int*)tmp.p[2], softmax, temperature, probs_dev,
                            (int*)tmp.p[0], (float*)tmp.p[1], s, out);
}

int krk_upsample_sigmoid(const float* x_dev, int C, int h, int w, int H, int W, float* y_dev, void* stream) {
    if (!x_dev || !y_dev || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return fail(KRK_E_INVALID, "krk_upsample_sigmoid: bad argument");
    if (krk_device_count() <= 0) return fail(KRK_E_HIP, "krk_upsample_sigmoid: no HIP device");
    if (krk_launch_upsample_sigmoid(x_dev, y_dev, C, h, w, H, W, (hipStream_t)stream))
        return fail(KRK_E_HIP, std::string("krk_upsample_sigmoid: launch failed: ") + hipGetErrorString(hipGetLastError()));
    return KRK_OK;
}

// forward declaration: the v / 255 table shared by the preprocessing entry points
static int prep_lut(const char* who, float** out);

int krk_dewarp_measure(const unsigned char* crops_dev, const int* desc_dev, int n, int max_w, int max_h, const double* weights_dev,
                       double* scratch_dev, int* work_dev, int* info_dev, void* stream) {
    return krk_dewarp_measure_page(crops_dev, 0, 1, desc_dev, n, max_w, max_h, weights_dev, scratch_dev, work_dev, info_dev, stream);
}

int krk_dewarp_apply(const unsigned char* crops_dev, const int* desc_dev, int n, int max_w, const int* work_dev, const int* geo_dev,
                     int out_h, int pad, int batch_w, float* x_dev, int* flags_dev, void* stream) {
    return krk_dewarp_apply_page(crops_dev, 0, 1, desc_dev, n, max_w, work_dev, geo_dev, out_h, pad, batch_w, x_dev, flags_dev, stream);
}

int krk_dewarp_measure_page(const unsigned char* crops_dev, long row_stride, int pix_stride, const int* desc_dev, int n, int max_w, int max_h,
                            const double* weights_dev, double* scratch_dev, int* work_dev, int* info_dev, void* stream) {
    if (!crops_dev || !desc_dev || !weights_dev || !scratch_dev || !work_dev || !info_dev || n < 0 || max_w <= 0 || max_h < 2 ||
        row_stride < 0 || (row_stride == 0 && pix_stride != 1))
        return fail(KRK_E_INVALID, "krk_dewarp_measure: bad argument");
    if (krk_device_count() <= 0) return fail(KRK_E_HIP, "krk_dewarp_measure: no HIP device");
    int* mm = work_dev;
    int* ridge = work_dev + 2 * (size_t)n;
    int* centre = ridge + (size_t)n * max_w;
    const int rc = krk_launch_dewarp_measure(crops_dev, (size_t)row_stride, pix_stride, desc_dev, n, max_w, max_h, weights_dev, scratch_dev, mm,
                                             ridge, centre, info_dev, (hipStream_t)stream);
    if (rc == -4) return fail(KRK_E_UNSUPPORTED, "krk_dewarp_measure: pixel stride must be 1, 3 or 4, lines at most 192 rows high");
    if (rc) return fail(KRK_E_HIP, std::string("krk_dewarp_measure: launch failed: ") + hipGetErrorString(hipGetLastError()));
    return KRK_OK;
}

int krk_dewarp_apply_page(const unsigned char* crops_dev, long row_stride, int pix_stride, const int* desc_dev, int n, int max_w,
                          const int* work_dev, const int* geo_dev, int out_h, int pad, int batch_w, float* x_dev, int* flags_dev, void* stream) {
    if (!crops_dev || !desc_dev || !work_dev || !geo_dev || !x_dev || !flags_dev || n < 0 || batch_w <= 0 || row_stride < 0 ||
        (row_stride == 0 && pix_stride != 1))
        return fail(KRK_E_INVALID, "krk_dewarp_apply: bad argument");
    if (krk_device_count() <= 0) return fail(KRK_E_HIP, "krk_dewarp_apply: no HIP device");
    float* lut = nullptr;
    if (int rc = prep_lut("krk_dewarp_apply", &lut)) return rc;
    const int* mm = work_dev;
    const int* centre = work_dev + 2 * (size_t)n + (size_t)n * max_w;
    const int rc = krk_launch_dewarp_apply(crops_dev, (size_t)row_stride, pix_stride, desc_dev, n, max_w, mm, centre, geo_dev, lut, out_h, pad,
                                           batch_w, x_dev, flags_dev, (hipStream_t)stream);
    if (rc == -4) return fail(KRK_E_UNSUPPORTED, "krk_dewarp_apply: needs pad > 0, out_h > 0 and a pixel stride of 1, 3 or 4");
    if (rc) return fail(KRK_E_HIP, std::string("krk_dewarp_apply: launch failed: ") + hipGetErrorString(hipGetLastError()));
    return KRK_OK;
}

// uint8 -> float table of ToDtype(scale=True): v / 255 in fp32, one per device
static int prep_lut(const char* who, float** out) {
    static float* luts[64] = {nullptr};
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(KRK_E_INVALID, std::string(who) + ": device index");
    if (!luts[dev]) {
        float h[256];
        for (int v = 0; v < 256; ++v) h[v] = (float)v / 255.0f;
        HIPCHK(hipMalloc((void**)&luts[dev], sizeof(h)));
        HIPCHK(hipMemcpy(luts[dev], h, sizeof(h), hipMemcpyHostToDevice));
    }
    *out = luts[dev];
    return KRK_OK;
}

int krk_prep_lines(const unsigned char* page_dev, int page_h, int page_w, int channels, const int* boxes_dev, int n,
                   int max_in_h, int out_h, int pad, int batch_w, float* x_dev, int* flags_dev, void* stream) {
    return krk_prep_lines_fmt(page_dev, page_h, page_w, (long)page_w * channels, channels, channels, boxes_dev, n, max_in_h, out_h, pad, batch_w,
                              x_dev, flags_dev, stream);
}

int krk_prep_lines_fmt(const unsigned char* page_dev, int page_h, int page_w, long row_stride, int pix_stride, int channels,
                       const int* boxes_dev, int n, int max_in_h, int out_h, int pad, int batch_w, float* x_dev, int* flags_dev, void* stream) {
    if (!page_dev || !boxes_dev || !x_dev || !flags_dev || n < 0 || batch_w <= 0 || row_stride <= 0)
        return fail(KRK_E_INVALID, "krk_prep_lines: bad argument");
    if (krk_device_count() <= 0) return fail(KRK_E_HIP, "krk_prep_lines: no HIP device");
    float* lut = nullptr;
    if (int rc = prep_lut("krk_prep_lines", &lut)) return rc;
    const int rc = krk_launch_prep_lines(page_dev, page_h, page_w, (size_t)row_stride, pix_stride, channels, boxes_dev, n, max_in_h, lut, out_h,
                                         pad, batch_w, x_dev, flags_dev, (hipStream_t)stream);
    if (rc == -4)
        return fail(KRK_E_UNSUPPORTED, "krk_prep_lines: line geometry outside the kernel's range (height, scale, padding) or a page format "
                                       "it does not read (pixel stride)");
    if (rc) return fail(KRK_E_HIP, std::string("krk_prep_lines: launch failed: ") + hipGetErrorString(hipGetLastError()));
    return KRK_OK;
}

int krk_prep_crops(const unsigned char* crops_dev, int channels, const int* desc_dev, int n, int max_in_h, int out_h, int pad,
                   int batch_w, float* x_dev, int* flags_dev, void* stream) {
    if (!crops_dev || !desc_dev || !x_dev || !flags_dev || n < 0 || batch_w <= 0)
        return fail(KRK_E_INVALID, "krk_prep_crops: bad argument");
    if (krk_device_count() <= 0) return fail(KRK_E_HIP, "krk_prep_crops: no HIP device");
    float* lut = nullptr;
    if (int rc = prep_lut("krk_prep_crops", &lut)) return rc;
    const int rc = krk_launch_prep_crops(crops_dev, channels, desc_dev, n, max_in_h, lut, out_h, pad, batch_w, x_dev, flags_dev,
                                         (hipStream_t)stream);
    if (rc == -4) return fail(KRK_E_UNSUPPORTED, "krk_prep_crops: line geometry outside the kernel's range (height, scale, padding)");
    if (rc) return fail(KRK_E_HIP, std::string("krk_prep_crops: launch failed: ") + hipGetErrorString(hipGetLastError()));
    return KRK_OK;
}

int krk_recognize(krk_plan* plan, const float* x_dev, const int* lens_host, int N, int W, float temperature,
                  void* stream, float* logits_dev, float* probs_dev, int* olens_host,
                  const krk_decode_out* out) {
    if (!plan || plan->steps.empty()) return fail(KRK_E_INVALID, "krk_recognize: null plan");
    const Step& last = plan->steps.back();
    if (last.kind != S_LINEAR) return fail(KRK_E_UNSUPPORTED, "krk_recognize: the network must end in a linear (O1) layer");
    if (plan->batch_ops)
        return fail(KRK_E_UNSUPPORTED, "krk_recognize: the network changes the number of lines (Addition / Reshape on the batch axis): "
                                       "its output lines are not the caller's lines; krk_forward + krk_greedy_decode");
    hipStream_t s = (hipStream_t)stream;
    RoctxRange whole(roctx().on() ? "krk_recognize N=" + std::to_string(N) + " W=" + std::to_string(W) : std::string());
    const float* logits = nullptr;
    const int* d_olens = nullptr;
    int T = 0;
    int rc = run_plan(plan, x_dev, lens_host, N, W, s, logits_dev, &logits, &d_olens, &T);
    if (rc) return rc;
    RoctxRange dec("softmax + greedy decode");
    const int C = last.outC;
    const size_t rows = (size_t)N * T;
    if (plan->d_labels.ensure(rows * sizeof(int)) || plan->d_confs.ensure(rows * sizeof(float)))
        return fail(KRK_E_NOMEM, "krk_recognize: workspace allocation failed");
    rc = decode_on_device(logits, (long)T * C, 1, C, N, C, T, d_olens, 1, temperature, probs_dev,
                          (int*)plan->d_labels.p, (float*)plan->d_confs.p, s, out);
    if (rc) return rc;
    if (olens_host) {
        // with the batch's width: a general Reshape scales seq_lens by it (krk_plan_olens refuses such plans and writes nothing)
        if (lens_host) {
            if ((rc = krk_plan_olens_w(plan, lens_host, N, W, olens_host))) return rc;
        } else
            for (int n = 0; n < N; ++n) olens_host[n] = T;
    }
    return KRK_OK;
}

}  // extern "C"
