// Shared declarations of the gfx950 line-recognition kernels (host + device).
// Written for CDNA4 only: 64-wide wavefronts, v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32 (exact f32 matrix cores), 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Ablation probes (skip the K loop / the loads / the stores of a kernel to price its phases) exist only in the
// -DKRK_ABLATE build (python -m kraken_amd.build --ablate -> libkraken_amd_ablate.so, selected with KRAKEN_AMD_LIB);
// the release kernels carry no probe branches in their hot loops.
#ifdef KRK_ABLATE
#define KRK_DBGBIT(a, bit) (((a).dbg & (bit)) != 0)
#else
#define KRK_DBGBIT(a, bit) (false)
#endif

// The split-bf16 kernel sources are compiled twice: as written ("bf16x3": a_hi*b_hi + a_hi*b_lo + a_lo*b_hi) and with
// -DKRK_BF16_ONE, which drops the two cross terms (KRK_CROSS) and renames the launchers (KRK_FN: name_b1): the opt-in
// plain-bf16 plan (KRK_PREC_BF16), one MFMA per product.  Same layouts, same code path, a third of the matrix work.
#ifdef KRK_BF16_ONE
#define KRK_CROSS(...)
#define KRK_FN(name) name##_b1
#else
#define KRK_CROSS(...) __VA_ARGS__
#define KRK_FN(name) name
#endif

// Cycle accounting of a kernel's phases (dev tool, -DKRK_ABLATE builds with probe bit 64 only): every wave sums, per phase, the
// shader cycles between two stamps (s_memtime) and adds its totals to a per-kernel array of device counters at its end;
// tools/phase_stats.py reads them through krk_debug_phase_stats.  Compiled out of the release library.
#ifdef KRK_ABLATE
#define KRK_PHASES(n) unsigned long long ph_acc_[n] = {}; unsigned long long ph_t_ = 0
#define KRK_PH_START(a) do { if (KRK_DBGBIT(a, 64)) ph_t_ = __builtin_readcyclecounter(); } while (0)
#define KRK_PH(a, i) do { if (KRK_DBGBIT(a, 64)) { const unsigned long long n_ = __builtin_readcyclecounter(); ph_acc_[i] += n_ - ph_t_; ph_t_ = n_; } } while (0)
#define KRK_PH_FLUSH(a, arr, n) do { if (KRK_DBGBIT(a, 64) && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < (n); ++i_) atomicAdd(&(arr)[i_], ph_acc_[i_]); atomicAdd(&(arr)[n], 1ull); } } while (0)
#else
#define KRK_PHASES(n)
#define KRK_PH_START(a)
#define KRK_PH(a, i)
#define KRK_PH_FLUSH(a, arr, n)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- activations
#define ACT_LINEAR 0
#define ACT_RELU 1
#define ACT_TANH 2
#define ACT_LEAKY 3

__device__ __forceinline__ float krk_sigmoid(float x) {
    // 1 / (1 + e^-x); e^-x = 2^(-x log2 e) on v_exp_f32, v_rcp_f32 for the reciprocal
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float krk_tanh(float x) {
    // tanh x = 2 sigma(2x) - 1; saturates cleanly to +-1 (e^-2x -> inf / 0)
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f;
}
// One LSTM cell update from the pre-activations z = (i, f, g, o) of a unit: 5 v_exp + 2 v_rcp (the textbook form: 5 + 5).
//   c' = f c + i tanh(g) = [c (1+ei)(1+eg) + (1-eg)(1+ef)] / [(1+ei)(1+ef)(1+eg)],   ei = e^-zi, ef = e^-zf, eg = e^-2zg
//   h  = o tanh(c')      = (1 - et) / [(1+eo)(1+et)],                                 eo = e^-zo, et = e^-2c'
// Exponent arguments are clamped to +-20 (x log2 e): sigmoid / tanh are within 2e-9 of their limits there and the products
// of the denominators stay below 2^88.
__device__ __forceinline__ float krk_lstm_cell(const f32x4& z, float& c) {
    constexpr float L2E = 1.4426950408889634f, LIM = 28.853900817779268f;   // 20 log2 e
    const float ai = __builtin_amdgcn_fmed3f(-L2E * z[0], -LIM, LIM);
    const float af = __builtin_amdgcn_fmed3f(-L2E * z[1], -LIM, LIM);
    const float ag = __builtin_amdgcn_fmed3f(-2.f * L2E * z[2], -LIM, LIM);
    const float ao = __builtin_amdgcn_fmed3f(-L2E * z[3], -LIM, LIM);
    const float ei = __builtin_amdgcn_exp2f(ai), ef = __builtin_amdgcn_exp2f(af);
    const float eg = __builtin_amdgcn_exp2f(ag), eo = __builtin_amdgcn_exp2f(ao);
    const float pi = 1.f + ei, pf = 1.f + ef, pg = 1.f + eg, po = 1.f + eo;
    const float pig = pi * pg;
    const float num = __builtin_fmaf(c, pig, (1.f - eg) * pf);
    const float cn = num * __builtin_amdgcn_rcpf(pig * pf);
    c = cn;
    const float at = __builtin_amdgcn_fmed3f(-2.f * L2E * cn, -LIM, LIM);
    const float et = __builtin_amdgcn_exp2f(at);
    return (1.f - et) * __builtin_amdgcn_rcpf(po * (1.f + et));
}

__device__ __forceinline__ float krk_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_TANH: return tanhf(v);
        case ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        default: return v;
    }
}

// ------------------------------------------------------------------ conv/GEMM
// One kernel covers ActConv2D, the LSTM input projection and LinSoftmax: an
// implicit GEMM whose K axis enumerates (channel, dy, dx) through an LDS offset
// table, so any kernel size / stride / dilation / channel count runs on the
// matrix cores without materialising im2col.
struct ConvArgs {
    const float* x;       // IN_SEQ=0: (N,Cin,H,W) planes; IN_SEQ=1: [W][Cin] rows (N=1,H=1)
    float* y;             // see epilogues in conv_mfma.hip
    const float* wpack;   // [nchunks][KSGpad][CBpad][64][4] fragment order, 4 K-steps per lane contiguous
    const float* bias;    // [CBpad*32]
    const int* len_in;    // [N] valid input width per line or nullptr
    const int* len_out;   // [N] valid stored-output width per line or nullptr
    int N, Cin, H, W;
    int Cout, CBpad;
    int kh, kw, sh, sw, dh, dw, ph, pw;
    int Ho, Wo;           // conv output extent (before pooling)
    int Hy, Wy;           // stored extent (after the fused 2x2/2 max-pool, else == Ho,Wo)
    int act;
    int cchunk, nchunks, Kc;       // channels per LDS chunk, #chunks, K (= cchunk*kh*kw) per chunk
    int KSG, KSG_last, KSGpad;     // groups of 4 K-steps per chunk / in the last chunk / allocated (KSG + 3)
    int KS4;                       // offset-table entries per lane half (= 4 * KSGpad)
    int vec4;                      // IN_SEQ: rows are 16-byte loadable (Cin % 4 == 0, cchunk % 4 == 0)
    int IH, IW, RS, PS;   // staged tile rows/cols, row stride, plane stride (floats)
    int SR;               // 32-pixel segments per tile row: tile = (8/SR) rows x (32*SR) cols
    int tiles_h, tiles_w;
    int otab_floats;      // LDS floats reserved for the K-offset table
    // optional split-bf16 channels-last output (feeds conv_x3.hip): y_split = hi plane, lo at + y_plane
    void* y_split;
    size_t y_plane;
    long y_sn, y_sr, y_sc;
    long y_cs = 1;        // element stride between output channels: 1 = channels-last; the row pitch for the [N][H][C][pitch] planes the tap kernel reads
};

// ------------------------------------------------------- split-bf16 ("bf16x3") conv/GEMM
struct X3Args {
    const __bf16* x;      // hi plane of the split NHWC input [N][H][W][Cin]; lo plane at + x_plane elements
    size_t x_plane;
    void* y;              // OUT_F32: float rows; else hi plane of the split output, lo at + y_plane
    size_t y_plane;
    long y_sn, y_sr, y_sc;   // split output strides (elements): n, row, col (filter stride 1)
    const __bf16* wpack;  // [chunk][tap][KB][CBpad][plane][lane][8]
    const float* bias;    // [CBpad*32]
    const int* len_in;
    const int* len_out;
    int N, Cin, H, W;
    int Cout, CBpad;
    int kh, kw, sh, sw, dh, dw, ph, pw;
    int Ho, Wo, Hy, Wy;
    int act;
    int cchunk, nchunks, KB, KB_last;   // channels per LDS chunk (multiple of 16), 16-channel blocks per chunk
    int IH, IW, PSTR, lds_plane;        // LDS tile: pixels, bytes per pixel (padded), bytes per plane
    int y_f32;                   // 1: write fp32 NHWC (same element index as the hi plane) for a GroupNorm consumer
    int y_blkM, y_cols;          // > 0: sequence output in K-blocked order, y_blkM rows of y_cols per line (see gemm_x3.hip)
    int tps;                     // conv_x3p.hip: tile copies per weight-stage boundary (0: not eligible)
    int single_buf = 0;          // conv_x3p.hip: one tile buffer (three workgroups per CU), copies at the chunk boundaries
    int SR, tiles_h, tiles_w;    int dbg;                            // probe bits (env KRK_X3_DBG): 1 skip K loop, 2 skip staging loads, 4 skip stores
};

// first convolution of a one-channel image on the bf16 cores (conv1_x3.hip)
struct Conv1Args {
    const float* x;       // [N][Cin][H][W] fp32, Cin = 1 or 3
    const __bf16* wpack;  // [Cin][kh][plane][64 lanes][8]: lane (filter, half) holds taps 8*half..+7 of kernel row dy of a channel
    const float* bias;    // [32]
    __bf16* y;            // hi plane of the split channels-last output; lo plane at + y_plane
    size_t y_plane;
    long y_sn, y_sr, y_sc;
    const int* len_in;
    const int* len_out;
    int N, H, W, Cout, kh, kw, ph, pw;
    int Ho, Wo, Hy, Wy;
    int act, tiles_h, tiles_w;
    int dbg;              // probe bits (env KRK_X3_DBG): 1 no K loop, 2 no staging loads, 4 no stores
    int y_f32;            // 1: write fp32 NHWC instead of split planes (GroupNorm consumer)
    int y_pitch;          // > 0: write "NHCW" planes [N][Hy][Cout][y_pitch] (for conv_taps_x3.hip) instead of NHWC
    int Cin = 1;
};
bool krk_conv1_x3_supported(int Cin, int Cout, int kh, int kw, int sh, int sw, int dh, int dw);
int krk_launch_conv1_x3(const Conv1Args& a, bool pool, hipStream_t s);
int krk_launch_conv1_x3_b1(const Conv1Args& a, bool pool, hipStream_t s);

// wide-kernel convolution with taps as the K axis (conv_taps_x3.hip)
struct ConvTapArgs {
    const __bf16* x;      // hi plane [N][H][Cin][pitch]; lo plane at + x_plane elements
    size_t x_plane;
    const __bf16* wpack;  // [Cin][kh][plane][64 lanes][8]
    const __bf16* wpack5; // [Cin/2][kh][4 fragments][64 lanes][8]: the five-group packing (kw <= 13), or null
    const float* bias;    // [32]
    __bf16* y;            // split NHWC output (strides below), lo plane at + y_plane
    size_t y_plane;
    long y_sn, y_sr, y_sc;
    const int* len_out;
    int N, H, pitch, Cin, Cout, kh, kw, ph, pw;
    int Ho, Wo, Hy, Wy;
    int act, tiles_h, tiles_w;
    int y_f32;            // 1: write fp32 NHWC instead of split planes (GroupNorm consumer)
    int dbg;              // probe bits (env KRK_X3_DBG): 1 no K loop, 2 no weight refresh (FIVE), 4 no staging loads, 16 no stores
    int dma;              // 1: stage the input tile with raw-buffer -> LDS copies (both planes of a line within one 4 GB descriptor)
};
bool krk_conv_taps_supported(int Cin, int Cout, int kh, int kw, int sh, int sw, int dh, int dw);
int krk_launch_conv_taps(const ConvTapArgs& a, bool pool, hipStream_t s);
int krk_launch_conv_taps_b1(const ConvTapArgs& a, bool pool, hipStream_t s);

// split-bf16 row projection (gemm_x3.hip): Y[M][Cout] = X[M][K] . W^T + b
struct GemmX3Args {
    const __bf16* x;      // hi plane, K-blocked rows [K/8][M][8]; lo plane at + x_plane elements
    size_t x_plane;
    const __bf16* w;      // [ncg][K/16][plane][k-half][128 columns][8]
    const float* bias;    // [Cout]
    float* y;             // [M][Cout]
    int M, K, Cout;
    int ncg, ntiles;      // column groups of 128, row tiles of 256
    int tileT;            // > 0: rows are (line, step) with tileT steps per line; store row n*T+t at ((n/16)*T + t)*16 + n%16
                          // < 0: the INPUT rows are tile-time-major with -tileT steps per line: store row ((n/16)*T + t)*16 + n%16 at n*T+t
    int nlines;           // tileT < 0: lines that exist (rows of the padding lines of the last 16-line tile are dropped)
    int act;
    int dbg;              // probe bits (env KRK_X3_DBG): 1 no MFMA, 2 no copies, 4 no stores, 8 no LDS reads
    int nbuf;             // 3; 2 = probe (KRK_GEMM_SPREAD=0): all six copies of a K step in front of its MFMAs
};
int krk_launch_gemm_x3(const GemmX3Args& a, hipStream_t s);
int krk_launch_gemm_x3_b1(const GemmX3Args& a, hipStream_t s);

// ----------------------------------------------------------------------- LSTM
struct LstmArgs {
    const float* xp;      // [N*T][xstride] input projections (+ both biases), gate-interleaved columns
    const float* wp;      // [ndir][NG][NB][64][KG] recurrent weights, B-fragment order, KG K-steps per lane
    float* out;           // [N][T][ostride]
    const int* lens;      // [N] valid steps per line or nullptr
    int N, T, H, Hp;      // hidden size and hidden size padded to the column-block granule
    int NG, NB, G;        // K-groups (of KG MFMA K-steps) per time step, column blocks, G = 4*Hp gate columns/direction
    int ndir, dirmode;    // 1|2 directions; 0 fwd, 1 rev, 2 bidi
    int xstride, ostride;
    int dbg;              // probe bits (env KRK_LSTM_DBG): 1 no GEMM, 2 no gate math, 4 no output pass, 8 no x prefetch
    const float* peep = nullptr;   // lstm_big_kernel: [ndir][3][Hp] peephole weights of the ocropy cell (i, f, o), else null
    // lstm_big_kernel above the widths LDS holds (round 6): the cell state of a workgroup in HBM ([ndir][tiles][Hp][16 lines], Hp > 768)
    // and, above 1152, h as well ([ndir][tiles][parity 2][K rows][16 lines], zeroed by the host before the launch)
    float* cstate = nullptr;
    float* hstate = nullptr;
};

// small hidden sizes (Hp <= 32, lstm_small.hip): wp = [ndir][Hp/4 blocks][Hp/4 K steps][64 lanes]
bool krk_lstm_small_supported(int Hp);
int krk_launch_lstm_small(const LstmArgs& a, hipStream_t s);
// the same on the bf16 cores with split operands (bf16x3 plans): wx = [ndir][Hp/4 blocks][hi|lo][64 lanes][8] bf16, K slot (g, j) = unit 4 j + g
int krk_launch_lstm_small_x3(const LstmArgs& a, const void* wx, hipStream_t s);

// split-bf16 recurrent kernel (lstm_x3.hip), 16-line tiles
struct LstmX3Args {
    const float* xp;      // [N*T][xstride] fp32 input projections (+ biases), gate-interleaved columns
    const __bf16* wp;     // [ndir][NKB][NB][plane][lane][8] recurrent weights, split bf16, B-fragment order
    __bf16* out;          // hi plane, K-blocked sequence rows [ostride/8][N*T][8]; lo plane at + out_plane elements
    size_t out_plane;
    const int* lens;
    int N, T, H, Hp;
    int NKB, NB, G;       // K-blocks of 32, column blocks of 16, G = 4*Hp gate columns per direction
    int ndir, dirmode;
    int xstride, ostride;
    int hrow;             // bytes per LDS row of h (one line, one plane) = NKB*64 + 16
    int xtiled;           // 1: xp rows are tile-time-major (see gemm_x3.hip): row of (line n, step t) = ((n/16)*T + t)*16 + n%16
    int otiled;           // 1: the OUTPUT rows are tile-time-major too (ceil(N/16)*16*T rows per piece): the consumer is gemm_x3
    int dbg;              // probe bits (env KRK_LSTM_DBG): 1 no weight loads, 2 no gate math, 4 no MFMA, 8 no x prefetch
};
int krk_launch_lstm_x3(const LstmX3Args& a, hipStream_t s);

// weight-stationary cluster kernel (lstm_ws.hip): four workgroups hold W_hh in registers and exchange h_t through L2
struct LstmWsArgs {
    const float* xp;      // fp32 input projections (+ biases), gate-interleaved columns, rows TILE-TIME-MAJOR: (line n, step t) at ((n/16)*T + t)*16 + n%16
    const __bf16* wp;     // [ndir][slice 4][wave 8][BPW][NKB][plane][lane][8]: per-wave resident fragments, zero where a block does not exist
    __bf16* out;          // hi plane, K-blocked sequence rows [ostride/8][N*T][8]; lo plane at + out_plane elements
    size_t out_plane;
    const int* lens;
    int N, T, H, Hp;
    int NKB, NB, G;       // K-blocks of 32, gate-column blocks of 16, G = 4*Hp gate columns per direction
    int ndir, dirmode;
    int xstride, ostride;
    int hrow;             // bytes per LDS row of h (one line, one plane) = NKB*64 + 16
    int BPC;              // gate-column blocks per cluster slice = ceil(NB/4)
    unsigned long long* gran;   // exchange granules [cluster][group][parity 2][slice 4][BPC*4 units][16 lines], zeroed at allocation
    unsigned* ctrl;       // [0]: monotonic ticket counter (cluster membership is claimed at run time)
    unsigned ticket_base; // counter value before this launch (the host adds the grid size after every launch)
    unsigned epoch;       // launch number folded into the granule tags (never 0)
    unsigned* err;        // mapped host word: set to 1 if an exchange wait timed out
    int otiled;           // 1: output rows tile-time-major (ceil(N/16)*16*T rows per piece), whole 256-byte runs per (piece, step)
    int dbg;              // probe bits (-DKRK_ABLATE build, env KRK_LSTM_DBG): 1 no exchange reads, 2 no gate math / publish, 4 no MFMA, 8 no xproj loads, 16 no output pass, 32 no step barrier
};
bool krk_lstm_ws_supported(int H, int Hp);
int krk_lstm_ws_clusters(int N, int ndir, int groups);
size_t krk_lstm_ws_gran_bytes(int N, int ndir, int BPC, int groups);
int krk_launch_lstm_ws(const LstmWsArgs& a, int groups, hipStream_t s);
int krk_launch_lstm_ws_b1(const LstmWsArgs& a, int groups, hipStream_t s);


// K-steps whose B fragments one lane loads contiguously (dwordx4 granules) in the recurrent kernel
int krk_lstm_kg(int M, int blocks_per_wave);

// line preprocessing on the device (prep_lines.hip): boxes_dev = [n][5] int32 (x0, y0, x1, y1, out_w)
// page row y starts rs bytes behind row y - 1, a pixel is ps bytes (ps == ch: packed; 4: Pillow's RGBX; ch 1 of ps >= 3: 'L' conversion)
int krk_launch_prep_lines(const unsigned char* page, int page_h, int page_w, size_t rs, int ps, int ch, const int* boxes_dev, int n,
                          int max_in_h, const float* lut, int out_h, int pad, int batch_w, float* out, int* flags, hipStream_t s);

// the same for a packed buffer of uint8 line images: desc_dev = [n][4] int32 (byte offset, width, height, out_w)
int krk_launch_prep_crops(const unsigned char* crops, int ch, const int* desc_dev, int n, int max_in_h,
                          const float* lut, int out_h, int pad, int batch_w, float* out, int* flags, hipStream_t s);

// CenterNormalizer dewarp of 1-channel lines (dewarp.hip): measure (centre line, spread) and normalize + float stage
// rs = 0: packed line images (a row is the line's own width); rs > 0: crops of one page with rows rs bytes apart; ps = bytes per pixel
int krk_launch_dewarp_measure(const unsigned char* crops, size_t rs, int ps, const int* desc, int n, int maxw, int maxh, const double* wts,
                              double* scratch, int* mm, int* ridge, int* centre, int* info, hipStream_t s);
int krk_launch_dewarp_apply(const unsigned char* crops, size_t rs, int ps, const int* desc, int n, int maxw, const int* mm, const int* centre,
                            const int* geo, const float* lut, int out_h, int pad, int batch_w, float* out, int* flags, hipStream_t s);

// host-side launchers (implemented in the .hip files)
int krk_launch_conv(const ConvArgs& a, bool in_seq, bool out_seq, bool pool, hipStream_t s);
int krk_launch_lstm(const LstmArgs& a, int M, hipStream_t s);
int krk_launch_lstm_big(const LstmArgs& a, hipStream_t s);   // Hp > 256 (lstm_rec.hip); a.cstate above 768, a.hstate above 1152
size_t krk_lstm_big_cstate_floats(int N, int ndir, int Hp);   // 0: the cell state fits LDS
size_t krk_lstm_big_hstate_floats(int N, int ndir, int Hp);   // 0: h fits LDS
int krk_launch_conv_x3(const X3Args& a, bool out_f32, bool pool, hipStream_t s);
int krk_launch_conv_x3_b1(const X3Args& a, bool out_f32, bool pool, hipStream_t s);
// three-plane ("bf16x6") convolution in front of a GroupNorm (conv_x6.hip): fp32-class products on the bf16 cores, fp32 NCHW out
int krk_x6_cb(int Cout);
int krk_launch_conv_x6(const X3Args& a, bool pool, hipStream_t s);
// fp32 (N,C,H,W) -> three bf16 planes (h, m, l: x = h + m + l exactly to 2^-24) in NHWC order (norm_x3.hip)
int krk_launch_split3_nhwc(const float* x, void* y, size_t plane, int N, int C, int H, int W, hipStream_t s);
// pipelined variant (conv_x3p.hip): asynchronous double-buffered tile staging; split outputs only
int krk_conv_x3p_tps(int cchunk, int kb, int kb_last, int npix, int iw, int ntaps, int cout, size_t line_bytes);
int krk_launch_conv_x3p(const X3Args& a, bool pool, hipStream_t s);
int krk_launch_conv_x3p_b1(const X3Args& a, bool pool, hipStream_t s);
int krk_launch_split(const float* x, void* hi, size_t plane_elems, size_t n, hipStream_t s);
// fp32 rows [M][K] -> K-blocked split planes [K/8][M][8] (hi, lo at + M*K elements)
int krk_launch_split_rows(const float* x, void* hi, int M, int K, size_t plane, hipStream_t s);
int krk_x3_cb(int Cout);
int krk_launch_maxpool(const float* x, float* y, const int* len_out, int N, int C, int H, int W,
                       int kh, int kw, int sh, int sw, int Ho, int Wo, hipStream_t s);
// GroupNorm, optionally with the MaxPool that follows it (kh > 0: window kh x kw, stride sh x sw, output Ho x Wo, columns >=
// len_out[n] zero; kh == 0: plain GroupNorm, Ho / Wo ignored).  `scratch`: 2 * N*G * krk_groupnorm_chunks(...) doubles.
int krk_groupnorm_chunks(int N, int C, int H, int W, int G, int Ho);
int krk_launch_groupnorm(const float* x, float* y, const float* gamma, const float* beta, const int* lens, const int* len_out,
                         int N, int C, int H, int W, int G, float eps, int kh, int kw, int sh, int sw, int Ho, int Wo,
                         double* scratch, hipStream_t s);
// first convolution of a one-channel image fused into the GroupNorm (+ 2x2 MaxPool) behind it (c1gn.hip)
struct C1GnArgs {
    const float* x;        // (N, 1, H, W)
    const float* w;        // [C][9] filter taps (ky, kx), [C] bias
    const float* bias;
    const float* gamma;    // [C]
    const float* beta;
    const int* lens;       // [N] valid input (= convolution output) width, or null
    const int* len_out;    // [N] valid pooled width, or null
    double* part;          // [N * G][chunks][2]
    float* y;              // (N, C, Ho, Wo) pooled, or (N, C, H, W) without a pool
    int N, C, H, W, G, chunks, act, pool;
    int Ho, Wo;
    float eps;
};
bool krk_c1gn_supported(int Cin, int C, int kh, int kw, int sh, int sw, int dh, int dw, int G, int pool_kh, int pool_kw, int pool_sh,
                        int pool_sw);
int krk_c1gn_chunks(int N, int H);
int krk_launch_c1gn(const C1GnArgs& a, hipStream_t s);
int krk_launch_to_seq(const float* x, float* y, int N, int C, int H, int W, hipStream_t s);
// split-bf16 NHWC planes (norm_x3.hip): MaxPool, height collapse
int krk_launch_maxpool_x3(const void* x, size_t xplane, void* y, size_t yplane, const int* len_out, int N, int C, int H, int W,
                          int kh, int kw, int sh, int sw, int Ho, int Wo, hipStream_t s);
int krk_launch_toseq_x3(const void* x, void* y, size_t plane, int N, int C, int H, int W, hipStream_t s);
// fp32 (N,C,H,W) -> K-blocked split sequence rows in one pass (height collapse + split; norm_x3.hip)
int krk_launch_toseq_split_f32(const float* x, void* y, size_t plane, int N, int C, int H, int W, hipStream_t s);
// split NHWC planes (hi, lo at + plane elements) -> fp32 (N,C,H,W): where a bf16x3 plan continues with f32-only layers
int krk_launch_unsplit(const void* x, size_t plane, float* y, int N, int C, int H, int W, hipStream_t s);
// (N,C,H,W) <-> sequence rows [(n,h)][w][C] (yaxis = 0) or [(n,w)][h][C] (yaxis = 1) for LSTMs over image rows/columns
int krk_launch_img2rows(const float* x, float* y, int N, int C, int H, int W, int yaxis, hipStream_t s);
int krk_launch_rows2img(const float* x, float* y, int N, int C, int H, int W, int yaxis, const int* lens, int last, hipStream_t s);
// torch.cat on the channel axis, one source at a time, and Addition's sum over pieces of an axis (parallel groups, `A` layers)
int krk_launch_concat(const float* x, float* y, size_t outer, size_t inner, size_t stride, size_t off, hipStream_t s);
int krk_launch_chunk_sum(const float* x, float* y, size_t outer, size_t inner, int nk, size_t in_stride, hipStream_t s);
// general Reshape: y contiguous over the 5-D dims, y[c0..c4] = x[sum_i c_i * strides[i]]
int krk_launch_permute5(const float* x, float* y, const int dims[5], const size_t strides[5], hipStream_t s);
// zero insertion in front of a transposed convolution: y[p][y * sh][x * sw] = x[p][y][x], zeros elsewhere; planes = N * C
int krk_launch_upzero(const float* x, float* y, size_t planes, int H, int W, int sh, int sw, int Ho, int Wo, hipStream_t s);
// Softmax over the channels of an NCHW tensor (channel-softmax convolutions, the softmax heatmap head)
int krk_launch_softmax_c(const float* x, float* y, int N, int C, int H, int W, const int* lens, hipStream_t s);
// nearest upsampling + sigmoid of the segmenter's heatmaps: (C, h, w) -> (C, H, W)
int krk_launch_upsample_sigmoid(const float* x, float* y, int C, int h, int w, int H, int W, hipStream_t s);
int krk_launch_rowmax(const float* scores, long sn, long sc, long st, int N, int C, int T,
                      int softmax, float temp, float* probs, int* labels, float* confs,
                      hipStream_t s);
int krk_launch_collapse(const int* labels, const float* confs, const int* olens, int N, int T,
                        int* o_labels, int* o_starts, int* o_ends, float* o_confs, int* o_counts,
                        int t_stride, hipStream_t s);
