// Line preprocessing on the device for the rectangular-crop / fixed-height case: page (uint8, uploaded once) ->
// padded, inverted float line batch, written straight into the staging tensor the recognition plan reads.
// Replaces, bit for bit, the host chain the reference runs per line before the network
//   extract_polygons bbox branch: im.crop(bbox)                      kraken/lib/segmentation.py:1630-1643
//   ImageInputTransforms: fixed-height LANCZOS resize, v2.Pad(fill=255),
//                         PILToTensor, ToDtype(scale=True), max - x    kraken/lib/dataset/utils.py:93-152,
//                                                                     kraken/lib/functional_im_transforms.py:58-82
// whose arithmetic lives in Pillow (un-vendored; `_fixed_resize` -> Image.resize(LANCZOS) -> ImagingResample, 8 bits per
// channel).  Pillow's published algorithm, restated here:
//   * separable two-pass resampling, horizontal first (only the source rows the vertical pass will read), each pass
//     storing clipped uint8;
//   * per output index xx: scale = in/out, filterscale = max(scale, 1), support = 3 * filterscale,
//     center = (xx + 0.5) * scale, window [int(center - support + 0.5), int(center + support + 0.5)) clipped to the
//     image, weights lanczos((x - center + 0.5) / filterscale) = sinc(t) sinc(t/3) normalised to sum 1 in double, then
//     rounded to 22-bit fixed point (int)(w * 2^22 +- 0.5);
//   * pixel = clip8((2^21 + sum pixel_i * k_i) >> 22).
// The weights are recomputed here in fp64 by every thread that needs them (a window is <= 2*ceil(3*scale)+1 taps): a
// 256-line batch would otherwise ship 30 MB of coefficient tables over PCIe for 14 MB of pixels.
// The float stage: ToDtype(scale=True) is uint8 / 255 in fp32 (a 256-entry table made on the host by the same division),
// `max - x` with max = 1.0 because the white padding is part of the tensor (pad > 0 is required; pad == 0 stays on the host).
// One workgroup = 64 output columns of one line (all out_h <= 128 rows of them).  Integer / byte work; round 6: 0.37 -> 0.18 ms per 256 lines
// at 72 -> 48 rows (profiles/r06_prep_kernels.txt: weight-table rows sized by the launch, odd pitch, 16-byte pixel loads).
#include "common.h"

namespace {

// (Pillow's C code rounds every product and sum; no fused multiply-adds in the weights -- see dewarp.hip for what contraction cost there)
#pragma clang fp contract(off)
typedef unsigned wide16 __attribute__((ext_vector_type(4), aligned(1)));      // 16 bytes at any address (global memory: unaligned access is on)
constexpr int PREC_BITS = 32 - 8 - 2;
constexpr int COLS = 64;          // output columns per workgroup
constexpr int MAX_OUT_H = 128;    // model input heights up to 128 (kraken's default recognition spec is 120 high; round 6: was 64)
constexpr int MAX_K = 96;         // taps per output sample (scale up to ~15)
// Weight-table rows in LDS are as long as the launch needs them (vertical: the tallest crop's scale; horizontal: that + slack, see the kernel),
// always an ODD number of ints: at the fixed 96 of rounds 3-5 the 64 lanes of the horizontal pass -- one sample each -- hit two banks on
// every weight read and write, and 44 of the 47 KB a workgroup asked for were table rows nobody wrote (three workgroups per CU).
constexpr int MAX_ROWS = 768;     // source rows of a line

__device__ __forceinline__ double sinc_f(double x) {
    if (x == 0.0) return 1.0;
    x = x * 3.14159265358979323846;
    return sin(x) / x;
}
__device__ __forceinline__ double lanczos_f(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_f(x) * sinc_f(x / 3);
    return 0.0;
}

// window + fixed-point weights of output sample xx (Pillow precompute_coeffs + normalize_coeffs_8bpc); returns xmax
__device__ int resample_weights(int in_size, int out_size, int xx, int* xmin_out, int* k /* [cap] */, int cap = MAX_K) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > cap) xmax = cap;              // the launcher rejects such scales; keeps the loops (and the table rows) bounded
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += lanczos_f((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
        double w = lanczos_f((x + xmin - center + 0.5) * ss);
        if (ww != 0.0) w /= ww;
        k[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << PREC_BITS)) : (int)(0.5 + w * (double)(1 << PREC_BITS));
    }
    *xmin_out = xmin;
    return xmax;
}

__device__ __forceinline__ unsigned clip8(int v) {
    v >>= PREC_BITS;
    return v < 0 ? 0u : (v > 255 ? 255u : (unsigned)v);
}

// PACKED = false: crops of ONE page, boxes [n][5] = x0 y0 x1 y1 out_w (krk_prep_lines).
// PACKED = true: every line brings its own crop -- `page` is a packed buffer of uint8 images, boxes [n][4] = byte offset,
//                width, height, out_w (krk_prep_crops): what a host-side line extractor (baseline / polygon extraction, any
//                producer of line images) hands over, 1 byte per pixel over PCIe instead of the 4 of a float tensor.
// Page pixels: row y starts `rs` bytes after row y - 1, a pixel is `ps` bytes.  ps == ch: packed channels (what np.asarray(im)
// gives); ps == 4 with ch == 3: Pillow's own storage of an 'RGB' image (R, G, B, X: the rows travel as they lie in Pillow's
// memory); ch == 1 with ps >= 3: the 1-channel model reads Pillow's 'L' conversion of the colour pixel,
// (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16 (libImaging/Convert.c, bit for bit: tests/test_gpu_parity.py).
template <bool PACKED>
__global__ void __launch_bounds__(256) prep_lines_kernel(const unsigned char* __restrict__ page, int page_h, int page_w, int ch,
                                                         size_t rs, int ps,
                                                         const int* __restrict__ boxes,
                                                         const float* __restrict__ lut, int out_h, int pad, int batch_w,
                                                         float* __restrict__ out, int* __restrict__ flags, int kvs, int khs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = blockIdx.y;
    int x0, y0, in_w, in_h, ow;
    if constexpr (PACKED) {
        const int* b = boxes + 4 * n;
        page += (size_t)(unsigned)b[0];
        x0 = 0; y0 = 0; in_w = b[1]; in_h = b[2]; ow = b[3];
        page_w = in_w; page_h = in_h;
        ps = ch; rs = (size_t)in_w * ch;
    } else {
        const int* b = boxes + 5 * n;
        x0 = b[0]; y0 = b[1]; in_w = b[2] - b[0]; in_h = b[3] - b[1]; ow = b[4];
    }
    const int col0 = blockIdx.x * COLS;             // first column of this tile in the PADDED line [0, ow + 2*pad)
    const int line_w = ow + 2 * pad;
    const int tid = threadIdx.x;
    float* orow = out + (size_t)n * ch * out_h * batch_w;
    if (col0 >= line_w || in_w <= 0 || in_h <= 0 || ow <= 0) {
        // batch padding right of the line (or an empty line): zeros
        for (int e = tid; e < ch * out_h * COLS; e += 256) {
            const int c = e / (out_h * COLS), r = (e / COLS) % out_h, x = col0 + e % COLS;
            if (x < batch_w) orow[((size_t)c * out_h + r) * batch_w + x] = 0.f;
        }
        return;
    }
    // vertical pass geometry: source rows [yfirst, ylast) feed it (Pillow: ybox_first / ybox_last)
    // `khs` = ints per row of the HORIZONTAL table as the launcher sized it (the vertical scale's taps + slack: a crop is scaled by the same
    // factor in both directions, up to the rounding of its output width).  A line that needs more per column (a crop a few pixels wide:
    // out_w = int(w * out_h / h) rounds its scale up a lot) gets fewer columns per trip through the table instead: `hcols` of the 64.
    int KSTR = khs, hcols = COLS;
    {
        const double hscale = (double)in_w / (double)max(ow, 1);
        const int need = min(MAX_K, (int)(6.0 * (hscale < 1.0 ? 1.0 : hscale)) + 3) | 1;
        if (need > khs) { KSTR = need; hcols = max(1, COLS * khs / need); }
    }
    // `kvs` = ints per row of the VERTICAL table: the launcher knows the tallest crop, so the rows hold the taps of THAT scale, not 96
    // (round 6: 18.8 of the 47 KB a workgroup asked for at 72 -> 48 rows were table rows nobody wrote: three workgroups per CU, two for colour)
    int* kv = reinterpret_cast<int*>(smem);                          // [out_h][kvs]
    int* kvb = kv + out_h * kvs;                                    // [out_h][2]
    int* kh = kvb + out_h * 2;                                       // [hcols][KSTR] inside COLS * khs ints
    int* khb = kh + COLS * khs;                                     // [COLS][2]
    unsigned char* tmp = reinterpret_cast<unsigned char*>(khb + COLS * 2);   // [ch][rows][COLS]
    if (tid < out_h) {
        int xmin;
        const int xmax = resample_weights(in_h, out_h, tid, &xmin, kv + tid * kvs, kvs);
        kvb[2 * tid] = xmin;
        kvb[2 * tid + 1] = xmax;
    }
    const bool luma = ch == 1 && ps >= 3;
    int yfirst = 0, rows = 0;
    for (int cb0 = 0; cb0 < COLS; cb0 += hcols) {                // (one trip unless the line is a sliver: see hcols)
    const int nbk = min(hcols, COLS - cb0);
    if (cb0) __syncthreads();                                    // the table is reused: everyone is done with the previous columns
    if (tid >= MAX_OUT_H && tid < MAX_OUT_H + nbk) {             // (threads 0 .. out_h - 1 make the vertical weights, 128 .. 191 the horizontal ones)
        const int jj = tid - MAX_OUT_H, j = cb0 + jj, xx = col0 + j - pad;
        int xmin = 0, xmax = 0;
        if (xx >= 0 && xx < ow) xmax = resample_weights(in_w, ow, xx, &xmin, kh + jj * KSTR, KSTR);
        khb[2 * j] = xmin;
        khb[2 * j + 1] = xmax;
    }
    __syncthreads();
    yfirst = kvb[0];
    rows = kvb[2 * (out_h - 1)] + kvb[2 * (out_h - 1) + 1] - yfirst;
    // horizontal pass: tmp[c][r][j] for the source rows the vertical pass reads.  One thread per (row, column) does ALL channels:
    // the tap loop, its bounds tests and the weight reads are shared by the channels (they were repeated per channel), and a
    // 4-byte pixel is one load
    for (int e = tid; e < rows * nbk; e += 256) {
        int jj, r;
        if (nbk == COLS) { jj = e % COLS; r = e / COLS; } else { r = e / nbk; jj = e - r * nbk; }
        const int j = cb0 + jj;
        const int xmin = khb[2 * j], xmax = khb[2 * j + 1];
        unsigned v0 = 255, v1 = 255, v2 = 255;
        if (xmax > 0) {
            const int gy = y0 + yfirst + r;
            int s0 = 1 << (PREC_BITS - 1), s1 = s0, s2 = s0;
            const int* k = kh + jj * KSTR;
            // Image.crop pads what lies outside the page with 0
            // Round 6: the window's pixels are consecutive bytes -- when it lies inside the page row they are fetched 16 bytes at a
            // time (one load for up to 16 one-byte pixels) instead of one load per tap (1-byte pixels; four-byte pixels gained nothing).
            const int g0 = x0 + xmin;
            if (gy >= 0 && gy < page_h && g0 >= 0 && g0 + xmax <= page_w && ps == 1) {
                const unsigned char* src = page + (size_t)gy * rs + (size_t)g0;
                int x = 0;
                for (; x + 16 <= xmax || (x < xmax && g0 + x + 16 <= page_w); x += 16) {         // 16 bytes that lie inside the row
                    const wide16 v = *reinterpret_cast<const wide16*>(src + x);
                    const int nb = min(16, xmax - x);
#pragma unroll
                    for (int b = 0; b < 16; ++b)
                        if (b < nb) s0 += (int)((v[b >> 2] >> (8 * (b & 3))) & 255u) * k[x + b];
                }
                for (; x < xmax; ++x) s0 += (int)src[x] * k[x];                                     // (a window that ends at the row's last bytes)
            } else if (gy >= 0 && gy < page_h) {
                const unsigned char* row = page + (size_t)gy * rs;
                for (int x = 0; x < xmax; ++x) {
                    const int gx = x0 + xmin + x;
                    if (gx < 0 || gx >= page_w) continue;
                    const int kx = k[x];
                    if (ps == 4) {
                        const unsigned px = *reinterpret_cast<const unsigned*>(row + (size_t)gx * 4);     // R | G << 8 | B << 16 | X << 24
                        const unsigned R = px & 255u, G = (px >> 8) & 255u, B = (px >> 16) & 255u;
                        if (luma) s0 += (int)((R * 19595u + G * 38470u + B * 7471u + 0x8000u) >> 16) * kx;
                        else { s0 += (int)R * kx; s1 += (int)G * kx; s2 += (int)B * kx; }
                    } else {
                        const unsigned char* q = row + (size_t)gx * ps;
                        if (luma) s0 += (int)((q[0] * 19595u + q[1] * 38470u + q[2] * 7471u + 0x8000u) >> 16) * kx;
                        else if (ch == 1) s0 += (int)q[0] * kx;
                        else { s0 += (int)q[0] * kx; s1 += (int)q[1] * kx; s2 += (int)q[2] * kx; }
                    }
                }
            }
            v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
        }
        tmp[(0 * rows + r) * COLS + j] = (unsigned char)v0;
        if (ch == 3) {
            tmp[(1 * rows + r) * COLS + j] = (unsigned char)v1;
            tmp[(2 * rows + r) * COLS + j] = (unsigned char)v2;
        }
    }
    }
    __syncthreads();
    // vertical pass + white padding + float + invert
    bool ink = false;
    for (int e = tid; e < ch * out_h * COLS; e += 256) {
        const int j = e % COLS, yy = (e / COLS) % out_h, c = e / (COLS * out_h);
        const int x = col0 + j;
        if (x >= batch_w) continue;
        float val = 0.f;                               // padding columns: 255 -> 1 - 1 = 0; right of the line: batch padding
        const int xx = x - pad;
        if (xx >= 0 && xx < ow) {
            const int ymin = kvb[2 * yy] - yfirst, ymax = kvb[2 * yy + 1];
            int ss0 = 1 << (PREC_BITS - 1);
            const int* k = kv + yy * kvs;
            for (int y = 0; y < ymax; ++y) ss0 += (int)tmp[(c * rows + ymin + y) * COLS + j] * k[y];
            const unsigned v = clip8(ss0);
            val = 1.0f - lut[v];
            ink = ink || v != 255u;
        }
        orow[((size_t)c * out_h + yy) * batch_w + x] = val;
    }
    if (__any(ink) && (tid & 63) == 0) atomicOr(flags + n, 1);
}

}  // namespace

// ints per row of the vertical weight table: the taps of the tallest crop's scale (window = [int(c - s + 0.5), int(c + s + 0.5)), s = 3 * scale:
// at most 6 * scale + 2 samples), odd (bank spread), never more than MAX_K (+ 1)
static int vertical_row_ints(int max_in_h, int out_h) {
    const double scale = std::max(1.0, (double)max_in_h / std::max(out_h, 1));
    const int taps = std::min(MAX_K, (int)(6.0 * scale) + 3);
    return taps | 1;
}

// LDS: out_h * (kvs + 2) + COLS * (khs + 2) ints + ch * rows * COLS bytes
int krk_launch_prep_lines(const unsigned char* page, int page_h, int page_w, size_t rs, int ps, int ch, const int* boxes_dev, int n,
                          int max_in_h, const float* lut, int out_h, int pad, int batch_w, float* out, int* flags, hipStream_t s) {
    if (n <= 0) return 0;
    if (out_h < 1 || out_h > MAX_OUT_H || pad < 1 || (ch != 1 && ch != 3) || max_in_h > MAX_ROWS) return -4;
    if (ps < ch || ps > 4 || (ch == 1 && ps == 2) || rs < (size_t)page_w * ps) return -4;
    const int kvs = vertical_row_ints(max_in_h, out_h);
    const int khs = std::min(MAX_K + 1, kvs + 2);                  // (odd like kvs)
    const size_t lds = ((size_t)out_h * (kvs + 2) + (size_t)COLS * (khs + 2)) * sizeof(int) + (size_t)ch * (max_in_h + 2) * COLS;
    if (lds > 160 * 1024) return -4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(prep_lines_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipMemsetAsync(flags, 0, (size_t)n * sizeof(int), s);
    dim3 grid((unsigned)((batch_w + COLS - 1) / COLS), (unsigned)n);
    hipLaunchKernelGGL(prep_lines_kernel<false>, grid, dim3(256), lds, s, page, page_h, page_w, ch, rs, ps, boxes_dev, lut, out_h, pad, batch_w,
                       out, flags, kvs, khs);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// packed crops: crops_dev = uint8 images back to back ([h][w][ch] each), desc_dev = [n][4] int32: byte offset, w, h, out_w
int krk_launch_prep_crops(const unsigned char* crops, int ch, const int* desc_dev, int n, int max_in_h,
                          const float* lut, int out_h, int pad, int batch_w, float* out, int* flags, hipStream_t s) {
    if (n <= 0) return 0;
    if (out_h < 1 || out_h > MAX_OUT_H || pad < 1 || (ch != 1 && ch != 3) || max_in_h > MAX_ROWS) return -4;
    const int kvs = vertical_row_ints(max_in_h, out_h);
    const int khs = std::min(MAX_K + 1, kvs + 2);                  // (odd like kvs)
    const size_t lds = ((size_t)out_h * (kvs + 2) + (size_t)COLS * (khs + 2)) * sizeof(int) + (size_t)ch * (max_in_h + 2) * COLS;
    if (lds > 160 * 1024) return -4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(prep_lines_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipMemsetAsync(flags, 0, (size_t)n * sizeof(int), s);
    dim3 grid((unsigned)((batch_w + COLS - 1) / COLS), (unsigned)n);
    hipLaunchKernelGGL(prep_lines_kernel<true>, grid, dim3(256), lds, s, crops, 0, 0, ch, (size_t)0, ch, desc_dev, lut, out_h, pad, batch_w, out,
                       flags, kvs, khs);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
