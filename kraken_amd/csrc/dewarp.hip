// CenterNormalizer dewarp of 1-channel bounding-box lines on the device (round 3).
// Replaces, bit for bit, what the reference runs per line on the host for a 1-channel model on a bbox segmentation:
//   functional_im_transforms.pil_dewarp -> lineest.dewarp / CenterNormalizer.measure + normalize   kraken/lib/lineest.py:26-87
// whose arithmetic lives in scipy.ndimage (un-vendored dependency; gaussian_filter, uniform_filter, affine_transform).  scipy's
// published algorithms are restated in oracle/np_oracle.py (center_normalize_np), pinned bit for bit against scipy itself and
// against the reference's outputs (tests/golden/transforms.npz); this file follows that restatement operation by operation in
// fp64 with explicit round-to-nearest multiplies / adds / divides (no FMA contraction), so the integer centre line -- an argmax
// followed by a truncation -- and therefore every output pixel is the reference's:
//   measure   ink = (top - v) / max ink;  blur = G_axis0(sigma h/2) then G_axis1(sigma h) (zero boundary; a symmetric kernel is
//             summed  w0 x[c] + sum_{j = r..1} w_j (x[c-j] + x[c+j]),  far to near);  blur += 0.001 * U(h/2 x w) (running sums:
//             t += x[l + size - 1] - x[l - 1]; out = t / size);  ridge = first argmax per column;  centre = trunc(G(sigma 0.3 h,
//             reflect boundary) of the integer ridge);  mad = mean |y - centre| over ink pixels (a sum of integers: exact);
//             r = int(1 + 4 mad)
//   normalize band[yy][x] = padded_line[centre[x] + h - r + yy][x], yy < 2r (float32);  output (target_h, int(scale * w)),
//             scale = target_h / 2r: bilinear at (y / scale, x / scale), paper (top) outside [0, n - 1], weights (wy wx) summed
//             over (0,0) (0,1) (1,0) (1,1) in fp64, stored as float32;  then the float stage of ImageInputTransforms: clip, uint8
//             truncation (array2pil), white padding, / 255, 1 - x
// Integer / fp64 work on a few MB per batch: latency- and HBM-bound, nowhere near a roofline that matters; what matters is that
// the 0.65 k lines/s of the scipy path (profiles/r02_bench_api.json) no longer feeds a 100 k lines/s recogniser.
// Per-line descriptor `desc` [n][8] int32: crop byte offset, w, h, scratch offset (doubles), weight-table offset (doubles),
// r0 (axis-0 radius), r1 (axis-1 radius), r2 (ridge radius); weight table of a height: [w0: 2 r0 + 1][w1: 2 r1 + 1][w2: 2 r2 + 1].
// Scratch per line: three planes of h * w doubles.  `info` [n][4] int32 out: r, ok, ink flag, unused; `centre` [n][maxw] int32.
#include "common.h"

// hipcc contracts a * b + c into one fused multiply-add by default (-ffp-contract=fast) -- also through HIP's __dmul_rn / __dadd_rn,
// which are plain operators compiled in the header's own context.  scipy rounds the product and the sum separately, and the last
// bit matters here (round 3: 4 of 60 random lines had a few columns' centre off by one row), so every multiply and add of this file
// is a plain operator under this pragma.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double dmul(double a, double b) { return a * b; }
__device__ __forceinline__ double dadd(double a, double b) { return a + b; }

// Where a line's pixels lie: packed images back to back (rs = 0: a row is the line's own w bytes, ps = 1) or crops of ONE uploaded
// page (rs = bytes per page row, `off` = byte offset of the crop's first pixel; ps = 1: an 'L' page, ps = 3 / 4: an RGB / RGBX page
// read through Pillow's 'L' conversion (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16 -- what im.crop(box).convert('L') holds).
struct Src { size_t rs; int ps; };
__device__ __forceinline__ int px_at(const unsigned char* p, int y, int x, int w, Src f) {
    const unsigned char* q = p + (size_t)y * (f.rs ? f.rs : (size_t)w) + (size_t)x * f.ps;
    return f.ps < 3 ? (int)q[0] : (int)((q[0] * 19595u + q[1] * 38470u + q[2] * 7471u + 0x8000u) >> 16);
}

struct LineD { int off, w, h, soff, woff, r0, r1, r2; };
__device__ __forceinline__ LineD line_of(const int* desc, int n) {
    const int* d = desc + 8 * n;
    return LineD{d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]};
}

// K0: top (max) and min of a line; one workgroup per line.  mm[n] = {top, min}
// (one workgroup per line -- a batch is 256 of them, one per CU -- so it is a wide one: 1024 threads keep 16 waves' loads in flight)
constexpr int DW_WIDE = 1024;
__global__ void __launch_bounds__(DW_WIDE) dw_minmax_kernel(const unsigned char* crops, Src f, const int* desc, int* mm) {
    const int n = blockIdx.x;
    const LineD L = line_of(desc, n);
    const unsigned char* p = crops + (size_t)(unsigned)L.off;
    int mx = 0, mn = 255;
    for (int y = threadIdx.x >> 6; y < L.h; y += DW_WIDE / 64)          // a wave per row: consecutive lanes, consecutive pixels
        for (int x = threadIdx.x & 63; x < L.w; x += 64) {
            const int v = px_at(p, y, x, L.w, f);
            mx = max(mx, v); mn = min(mn, v);
        }
    __shared__ int smx[DW_WIDE], smn[DW_WIDE];
    smx[threadIdx.x] = mx; smn[threadIdx.x] = mn;
    __syncthreads();
    for (int s = DW_WIDE / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { smx[threadIdx.x] = max(smx[threadIdx.x], smx[threadIdx.x + s]); smn[threadIdx.x] = min(smn[threadIdx.x], smn[threadIdx.x + s]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { mm[2 * n] = smx[0]; mm[2 * n + 1] = smn[0]; }
}

// K1: Gaussian along axis 0 (rows) of ink -> plane 0.  A workgroup takes 32 columns of one line: the ink values of the tile
// ((top - pixel) / amax, one fp64 DIVISION each) are computed once into LDS, with J + 3 rows of zeros above and below (scipy's zero
// boundary: no bounds test in the tap loop), and every thread then runs scipy's symmetric accumulation for FOUR ADJACENT ROWS of
// its column (round 6; before: one row per pass, two LDS reads per three fp64 operations -- LDS-bound at 0.27 / 0.70 ms per batch of
// 48- / 72-row lines): the left operand of output i at tap j is the left operand of output i + 1 at tap j + 1, so walking j downwards
// a thread needs one new value above and one below per tap for all four outputs -- 2 LDS reads for 12 fp64 operations, lanes on
// consecutive doubles.  Same operations in the same order per output as scipy's correlate1d (tests: bit-equal to scipy).
// Threads: 32 columns x (rows / 4) row blocks, up to 1024.  LDS: (h + 2 (J + 3)) x 32 doubles, J = min(r0, h) (+ the weights).
constexpr int G0_COLS = 32;
__global__ void __launch_bounds__(1024) dw_gauss0_kernel(const unsigned char* crops, Src f, const int* desc, const int* mm, const double* wts,
                                                         double* scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    const int n = blockIdx.y;
    const LineD L = line_of(desc, n);
    const int c = threadIdx.x & (G0_COLS - 1), g = threadIdx.x / G0_COLS, NG = blockDim.x / G0_COLS;
    const int x = blockIdx.x * G0_COLS + c;
    if (blockIdx.x * G0_COLS >= L.w) return;
    const double top = (double)mm[2 * n], amax = (double)(mm[2 * n] - mm[2 * n + 1]);
    if (amax == 0.0) return;
    const int J = min(L.r0, L.h);                    // beyond +-h both partners are outside: they add exactly 0
    const int P = J + 3, R = L.h + 2 * P;
    double* ink = reinterpret_cast<double*>(dw_smem);              // [R][32]: row P + y = ink row y, zeros outside
    double* wl = ink + (size_t)R * G0_COLS;                         // [J + 1]: w[r0 - j] at wl[j]
    const unsigned char* p = crops + (size_t)(unsigned)L.off;
    for (int r = g; r < R; r += NG) {
        const int y = r - P;
        ink[r * G0_COLS + c] = (y >= 0 && y < L.h && x < L.w) ? __ddiv_rn(dmul(top - (double)px_at(p, y, x, L.w, f), 1.0), amax) : 0.0;
    }
    const double* w = wts + L.woff;                  // w[r0 + j], j = -r0 .. r0
    for (int j = threadIdx.x; j <= J; j += blockDim.x) wl[j] = w[L.r0 - j];
    __syncthreads();
    if (x >= L.w) return;
    double* out = scratch + L.soff;
    const int nblk = (L.h + 3) / 4;
    for (int rb = g; rb < nblk; rb += NG) {
        const int y0 = 4 * rb;
        const double* ctr = ink + (size_t)(P + y0) * G0_COLS + c;          // ctr[k * 32] = ink row y0 + k
        double acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = dmul(ctr[i * G0_COLS], wl[0]);
        int j = J;
        for (; (j & 3) != 0; --j) {                                // the taps above the last multiple of 4
            const double wj = wl[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = dadd(acc[i], dmul(dadd(ctr[(i - j) * G0_COLS], ctr[(i + j) * G0_COLS]), wj));
        }
        if (j >= 4) {
            double lw[8], rw[8];                                   // lw[k] = row y0 + k - j, rw[k] = row y0 + k - 4 + j
#pragma unroll
            for (int i = 0; i < 4; ++i) { lw[i] = ctr[(i - j) * G0_COLS]; rw[4 + i] = ctr[(i + j) * G0_COLS]; }
            const double* lp = ctr + (4 - j) * G0_COLS;            // the next four rows above ...
            const double* rp = ctr + (j - 4) * G0_COLS;            // ... and below
#pragma unroll 2
            for (; j >= 4; j -= 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { lw[4 + i] = lp[i * G0_COLS]; rw[i] = rp[i * G0_COLS]; }
                lp += 4 * G0_COLS; rp -= 4 * G0_COLS;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {                   // taps j, j - 1, j - 2, j - 3
                    const double wj = wl[j - s4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = dadd(acc[i], dmul(dadd(lw[i + s4], rw[4 + i - s4]), wj));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { lw[i] = lw[4 + i]; rw[4 + i] = rw[i]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (y0 + i < L.h) out[(size_t)(y0 + i) * L.w + x] = acc[i];
    }
}

// K2: Gaussian along axis 1 (columns) plane 0 -> plane 1 (= blur).  sigma = h: 4 h taps to either side, ~390 for a 48-row line --
// the heaviest kernel of the measurement (2.8 ms per 256-line batch as one thread per output reading global memory).
// One wave takes 256 consecutive outputs of one row, FOUR ADJACENT ONES PER LANE: the left operand of output i at tap j is the
// left operand of output i + 1 at tap j + 1, so walking j downwards a lane needs one new value on the left and one on the right
// per tap for all four outputs -- 2 LDS reads for 12 fp64 operations.  The row segment and its halo (zeros outside the line) are
// staged in LDS split by index mod 4 (plane k holds the elements 4 q + k): the four values a lane fetches per group of four taps
// are then the SAME q in the four planes, lanes read consecutive doubles (no bank conflicts), and no index arithmetic depends on
// the tap.  The accumulation per output is scipy's correlate1d (symmetric case) operation for operation: centre tap first, then
// the pairs from the outermost inwards.
// (round 6: the plane pitch Q is a compile-time constant -- the four planes' reads of a step are ONE address with four immediate
// offsets; as a run-time value it cost eight pointer registers and their sixteen updates per eight taps.)
constexpr int G1_TILE = 256;
template <int Q>                                                   // doubles per plane: (G1_TILE + 2 * the largest halo of the launch) / 4
__global__ void __launch_bounds__(64) dw_gauss1_kernel(const int* desc, const int* mm, const double* wts, double* scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    const int n = blockIdx.z;
    const LineD L = line_of(desc, n);
    const int x0 = blockIdx.x * G1_TILE, y = blockIdx.y;
    if (x0 >= L.w || y >= L.h || mm[2 * n] == mm[2 * n + 1]) return;
    const int J = min(L.r1, L.w);
    const int Jp = (J + 3) & ~3;                                   // halo, a multiple of 4 elements
    double* pl = reinterpret_cast<double*>(dw_smem);              // [4][Q]: pl[k][q] = a[x0 - Jp + 4 q + k]
    double* wl = pl + 4 * Q;                                       // [J + 1]: w[r1 - j] at wl[j]
    const double* a = scratch + L.soff + (size_t)y * L.w;
    const double* w = wts + L.woff + (2 * L.r0 + 1);
    for (int e = threadIdx.x; e < G1_TILE + 2 * Jp; e += 64) {
        const int xx = x0 - Jp + e;
        pl[(e & 3) * Q + (e >> 2)] = (xx < 0 || xx >= L.w) ? 0.0 : a[xx];
    }
    for (int j = threadIdx.x; j <= J; j += 64) wl[j] = w[L.r1 - j];
    __syncthreads();
    const int t = threadIdx.x;
    const int qc = Jp / 4 + t;                                     // the lane's own four elements: pl[0..3][qc]
    auto el = [&](int rel) -> double {                             // element 4 t + rel of the tile (rel may be negative)
        const int e = Jp + 4 * t + rel;
        return pl[(e & 3) * Q + (e >> 2)];
    };
    double acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = dmul(pl[i * Q + qc], wl[0]);
    int j = J;
    for (; (j & 3) != 0; --j) {                                    // the taps above the last multiple of 4: generic reads
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = dadd(acc[i], dmul(dadd(el(i - j), el(i + j)), wl[j]));
    }
    if (j >= 4) {
        // j is a multiple of 4: left operands of the four outputs = plane i at q = qc - j/4, right ones at q = qc + j/4
        double lw[8], rw[8];                                       // lw[i] = element 4 t - j + i, rw[4 + i] = element 4 t + j + i
        int ql = qc - j / 4, qr = qc + j / 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) { lw[i] = pl[i * Q + ql]; rw[4 + i] = pl[i * Q + qr]; }
#pragma unroll 2
        for (; j >= 4; j -= 4) {
            ++ql; --qr;
#pragma unroll
            for (int i = 0; i < 4; ++i) { lw[4 + i] = pl[i * Q + ql]; rw[i] = pl[i * Q + qr]; }   // the next four on either side
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {                       // taps j, j - 1, j - 2, j - 3
                const double wj = wl[j - s4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = dadd(acc[i], dmul(dadd(lw[i + s4], rw[4 + i - s4]), wj));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { lw[i] = lw[4 + i]; rw[4 + i] = rw[i]; }
        }
    }
    double* out = scratch + L.soff + (size_t)L.h * L.w + (size_t)y * L.w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x0 + 4 * t + i;
        if (x < L.w) out[x] = acc[i];
    }
}

// K3: uniform filter along axis 0 (size int(h/2)) of blur (plane 1) -> plane 0; one thread per column, sequential like scipy
__global__ void __launch_bounds__(256) dw_unif0_kernel(const int* desc, const int* mm, double* scratch) {
    const int n = blockIdx.y;
    const LineD L = line_of(desc, n);
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= L.w || mm[2 * n] == mm[2 * n + 1]) return;
    const double* b = scratch + L.soff + (size_t)L.h * L.w;
    double* o = scratch + L.soff;
    const int size = (int)(L.h * 0.5), s1 = size / 2;
    auto ext = [&](int l) -> double { const int yy = l - s1; return (yy < 0 || yy >= L.h) ? 0.0 : b[(size_t)yy * L.w + x]; };
    const double dsize = (double)size;
    double t = 0.0;
    for (int l = 0; l < size; ++l) t = dadd(t, ext(l));
    o[x] = __ddiv_rn(t, dsize);
    for (int l = 1; l < L.h; ++l) {
        t = dadd(t, dadd(ext(l + size - 1), -ext(l - 1)));
        o[(size_t)l * L.w + x] = __ddiv_rn(t, dsize);
    }
}

// K4: uniform filter along axis 1 (size w) plane 0 -> plane 2: scipy's running sum per row, t += entering - leaving, which no
// parallel scan reproduces bit for bit -- a row is a sequential walk.  The window is as wide as the row, so a step never has both
// an entering and a leaving sample inside the row, and the walk is two passes over the row: pass 1 adds a[0 .. w-1] (the window
// fills, o[0] after a[last]; then samples enter: o[k - last] after a[k]), pass 2 subtracts a[0 .. w-2-s1] (o[k + s1 + 1] after a[k]).
// One wave walks 64 rows, a lane per row.  Lane-per-row global accesses touch 64 different cache lines per instruction and the
// walk then runs at one memory latency per 8 steps (0.6 ms per batch as first written); here the wave moves 64 x 32 tiles
// cooperatively -- coalesced 256-byte row pieces, the next tile's loads in flight while this one is walked from LDS -- and writes
// the outputs of a tile back the same way.  Arithmetic per step as scipy's (including the additions of +-0.0).
// Round 6: the walk is ONE wave per 64 rows, and that wave also did the fp64 DIVISION of every output (t / size: a dozen dependent
// instructions) -- 192 waves on a 1024-SIMD chip, 0.23-0.25 ms per batch.  Now a workgroup is eight waves: wave 0 walks (adds only,
// the running sums go to LDS), the other seven divide and store the tile walked one step earlier, and all eight fetch the next one.
constexpr int U1_COLS = 32, U1_PITCH = U1_COLS + 1, U1_THREADS = 512;
__global__ void __launch_bounds__(U1_THREADS) dw_unif1_kernel(const int* desc, const int* mm, double* scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    double* tin = reinterpret_cast<double*>(dw_smem);              // [2][64 * U1_PITCH]: the tile being walked, the next one
    double* tsum = tin + 2 * 64 * U1_PITCH;                         // [2][64 * U1_PITCH]: running sums of the tile walked now / before
    const int n = blockIdx.y;
    const LineD L = line_of(desc, n);
    const int y0 = blockIdx.x * 64;
    if (y0 >= L.h || mm[2 * n] == mm[2 * n + 1]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = min(64, L.h - y0);
    const double* a = scratch + L.soff + (size_t)y0 * L.w;
    double* o = scratch + L.soff + (size_t)2 * L.h * L.w + (size_t)y0 * L.w;
    const int w = L.w, size = w, s1 = size / 2, last = size - 1 - s1;
    const double dsize = (double)size;
    // the two passes as ONE stream of tiles: tiles [0, n1) cover a[0 .. w-1], tiles [n1, n1 + n2) cover a[0 .. w-2-s1]
    const int len2 = w - 1 - s1;                       // samples that leave (may be 0)
    const int n1 = (w + U1_COLS - 1) / U1_COLS, n2 = (len2 + U1_COLS - 1) / U1_COLS, nt = n1 + n2;
    constexpr int PER = 64 * U1_COLS / U1_THREADS;     // elements of a tile per thread (4)
    double reg[PER];
    auto fetch = [&](int tile) {                       // tile -> registers (zeros outside the pass / the rows): row e / 32, column e % 32
        const bool second = tile >= n1;
        const int c0 = (second ? tile - n1 : tile) * U1_COLS, lim = second ? len2 : w;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * U1_THREADS + tid, r = e / U1_COLS, c = c0 + (e % U1_COLS);
            reg[q] = (r < rows && c < lim) ? a[(size_t)r * w + c] : 0.0;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * U1_THREADS + tid;
            tin[buf * 64 * U1_PITCH + (e / U1_COLS) * U1_PITCH + (e % U1_COLS)] = reg[q];
        }
    };
    // geometry of a tile's outputs: pass 1: o[k - last] for k >= last; pass 2: o[k + s1 + 1]
    auto geom = [&](int tile, int& ob, int& skip, int& cnt) {
        const bool second = tile >= n1;
        const int c0 = (second ? tile - n1 : tile) * U1_COLS, lim = second ? len2 : w;
        cnt = min(U1_COLS, lim - c0);
        ob = second ? c0 + s1 + 1 : max(c0 - last, 0);
        skip = second ? 0 : max(last - c0, 0);         // leading samples of the tile that produce no output
    };
    auto divide = [&](int tile, int first, int stride) {           // running sums of `tile` -> outputs, by the threads first, first + stride, ...
        int ob, skip, cnt;
        geom(tile, ob, skip, cnt);
        const int nout = cnt - skip;
        const double* ts = tsum + (tile & 1) * 64 * U1_PITCH;
        for (int e = first; e < 64 * U1_COLS; e += stride) {
            const int r = e / U1_COLS, c = e % U1_COLS;
            if (r < rows && c < nout) o[(size_t)r * w + ob + c] = __ddiv_rn(ts[r * U1_PITCH + c], dsize);
        }
    };
    double t = 0.0;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int tile = 0; tile < nt; ++tile) {
        if (tile + 1 < nt) fetch(tile + 1);            // in flight while this tile is walked
        if (wave == 0) {
            int ob, skip, cnt;
            geom(tile, ob, skip, cnt);
            const double* mine = tin + (tile & 1) * 64 * U1_PITCH + lane * U1_PITCH;
            double* ts = tsum + (tile & 1) * 64 * U1_PITCH + lane * U1_PITCH;
            if (tile < n1) {
                for (int k = 0; k < cnt; ++k) {
                    t = dadd(t, mine[k]);
                    if (k >= skip) ts[k - skip] = t;
                }
            } else {
                for (int k = 0; k < cnt; ++k) {
                    t = dadd(t, dadd(0.0, -mine[k]));
                    ts[k] = t;
                }
            }
        } else if (tile > 0) {
            divide(tile - 1, tid - 64, U1_THREADS - 64);
        }
        __syncthreads();
        if (tile + 1 < nt) stage((tile + 1) & 1);
        __syncthreads();
    }
    divide(nt - 1, tid, U1_THREADS);
}

// K5: ridge[x] = first argmax over rows of blur + 0.001 * uniform
__global__ void __launch_bounds__(256) dw_ridge_kernel(const int* desc, const int* mm, const double* scratch, int* ridge, int maxw) {
    const int n = blockIdx.y;
    const LineD L = line_of(desc, n);
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= L.w || mm[2 * n] == mm[2 * n + 1]) return;
    const double* b = scratch + L.soff + (size_t)L.h * L.w;
    const double* u = scratch + L.soff + (size_t)2 * L.h * L.w;
    int best = 0;
    double bv = dadd(b[x], dmul(0.001, u[x]));
    for (int y = 1; y < L.h; ++y) {
        const double v = dadd(b[(size_t)y * L.w + x], dmul(0.001, u[(size_t)y * L.w + x]));
        if (v > bv) { bv = v; best = y; }
    }
    ridge[(size_t)n * maxw + x] = best;
}

// K6: centre = trunc(Gaussian(sigma 0.3 h, reflect) of the integer ridge)
__global__ void __launch_bounds__(256) dw_centre_kernel(const int* desc, const int* mm, const double* wts, const int* ridge, int* centre, int maxw) {
    const int n = blockIdx.y;
    const LineD L = line_of(desc, n);
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= L.w || mm[2 * n] == mm[2 * n + 1]) return;
    const int* rg = ridge + (size_t)n * maxw;
    const double* w = wts + L.woff + (2 * L.r0 + 1) + (2 * L.r1 + 1);
    auto at = [&](int xx) -> double {               // reflect: (d c b a | a b c d | d c b a), repeated
        int m = xx % (2 * L.w);
        if (m < 0) m += 2 * L.w;
        if (m >= L.w) m = 2 * L.w - 1 - m;
        return (double)rg[m];
    };
    double t = dmul((double)rg[x], w[L.r2]);
    for (int j = L.r2; j >= 1; --j)
        t = dadd(t, dmul(dadd(at(x - j), at(x + j)), w[L.r2 - j]));
    centre[(size_t)n * maxw + x] = (int)t;          // C truncation of the integer output array
}

// K7: r = int(1 + 4 * mean |y - centre| over ink pixels); band bounds.  info[n] = {r, ok, has ink, 0}; one workgroup per line
__global__ void __launch_bounds__(DW_WIDE) dw_spread_kernel(const unsigned char* crops, Src f, const int* desc, const int* mm, const int* centre,
                                                            int maxw, int* info) {
    const int n = blockIdx.x;
    const LineD L = line_of(desc, n);
    const int top = mm[2 * n];
    __shared__ long long ssum[DW_WIDE];
    __shared__ int scnt[DW_WIDE], smin[DW_WIDE], smax[DW_WIDE];
    long long sum = 0;                                  // integer sums: any order gives the same total
    int cnt = 0, cmin = 1 << 30, cmax = -(1 << 30);
    if (top != mm[2 * n + 1]) {
        const unsigned char* p = crops + (size_t)(unsigned)L.off;
        const int* c = centre + (size_t)n * maxw;
        for (int y = threadIdx.x >> 6; y < L.h; y += DW_WIDE / 64)
            for (int x = threadIdx.x & 63; x < L.w; x += 64)
                if (px_at(p, y, x, L.w, f) != top) { sum += abs(y - c[x]); ++cnt; }
        for (int x = threadIdx.x; x < L.w; x += DW_WIDE) { cmin = min(cmin, c[x]); cmax = max(cmax, c[x]); }
    }
    ssum[threadIdx.x] = sum; scnt[threadIdx.x] = cnt; smin[threadIdx.x] = cmin; smax[threadIdx.x] = cmax;
    __syncthreads();
    for (int s = DW_WIDE / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            ssum[threadIdx.x] += ssum[threadIdx.x + s]; scnt[threadIdx.x] += scnt[threadIdx.x + s];
            smin[threadIdx.x] = min(smin[threadIdx.x], smin[threadIdx.x + s]); smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int r = 0, ok = 0;
        if (scnt[0] > 0) {
            const double mad = __ddiv_rn((double)ssum[0], (double)scnt[0]);
            r = (int)dadd(1.0, dmul(4.0, mad));
            // the reference slices stack[mid - r : mid + r] of a (3h)-row stack: only full slices give a rectangular band
            ok = (r >= 1 && smin[0] + L.h - r >= 0 && smax[0] + L.h + r <= 3 * L.h) ? 1 : 0;
        }
        info[4 * n] = r; info[4 * n + 1] = ok; info[4 * n + 2] = top != mm[2 * n + 1]; info[4 * n + 3] = 0;
    }
}

// K8: normalize + the float stage.  geo [n][4] int32: r, out_w (int(scale * w)), use (0: leave zeros), 0.
// One thread per output COLUMN, walking down its out_h rows (round 6; before: one thread per output pixel -- every 256 pixels paid the two
// fp64 divisions of the zoom factor behind a barrier, every pixel its column's centre lookups and, if it was not white, an atomic on the
// line's flag: 0.52 ms per batch).  The column's x coordinate, its two centre-line entries and its weights are computed once; a wave writes
// 64 consecutive floats per row; the ink flag is one atomic per wave.  Arithmetic per pixel exactly as before (scipy's order).
__global__ void __launch_bounds__(256) dw_apply_kernel(const unsigned char* crops, Src f, const int* desc, const int* mm, const int* centre, int maxw,
                                                       const int* geo, const float* lut, int out_h, int pad, int batch_w, float* out, int* flags) {
    const int n = blockIdx.y;
    const LineD L = line_of(desc, n);
    const int X = blockIdx.x * 256 + threadIdx.x;
    const int r = geo[4 * n], ow = geo[4 * n + 1], use = geo[4 * n + 2];
    if (X >= batch_w) return;
    float* o = out + (size_t)n * out_h * batch_w + X;
    const int xx = X - pad;
    if (!use || xx < 0 || xx >= ow) {                               // white padding (1 - 255/255) and the batch padding right of the line
        for (int Y = 0; Y < out_h; ++Y) o[(size_t)Y * batch_w] = 0.f;
        return;
    }
    // the zoom factor is a per-line constant: two fp64 divisions per column (they were done per output pixel at first)
    const double scale = __ddiv_rn(dmul((double)out_h, 1.0), (double)max(2 * r, 1));
    const double z = __ddiv_rn(1.0, scale);
    const unsigned char* p = crops + (size_t)(unsigned)L.off;
    const int* c = centre + (size_t)n * maxw;
    const double top = (double)mm[2 * n];
    const int bh = 2 * r, bw = L.w;
    const double cx = dmul((double)xx, z);
    const bool x_in = !(cx < 0.0 || cx > (double)(bw - 1));
    const int x0 = x_in ? (int)floor(cx) : 0;
    const double tx = cx - (double)x0;
    const bool x1_in = x0 + 1 < bw;                                 // past the last sample: weight 0, value top
    const int base0 = c[x0] + L.h - r, base1 = x1_in ? c[x0 + 1] + L.h - r : 0;      // row of the (3h)-row stack at band row 0
    auto band = [&](int yy, int x, int base, bool xin) -> double {  // float32 of the padded line: exact for 8-bit values
        if (yy >= bh || !xin) return top;
        const int row = base + yy;
        return (row >= L.h && row < 2 * L.h) ? (double)px_at(p, row - L.h, x, L.w, f) : top;
    };
    bool ink = false;
    for (int Y = 0; Y < out_h; ++Y) {
        const double cy = dmul((double)Y, z);
        float val;
        if (cy < 0.0 || cy > (double)(bh - 1) || !x_in) {
            val = (float)top;
        } else {
            const int y0 = (int)floor(cy);
            const double ty = cy - (double)y0;
            double acc = dmul(band(y0, x0, base0, true), dmul(1.0 - ty, 1.0 - tx));
            acc = dadd(acc, dmul(band(y0, x0 + 1, base1, x1_in), dmul(1.0 - ty, tx)));
            acc = dadd(acc, dmul(band(y0 + 1, x0, base0, true), dmul(ty, 1.0 - tx)));
            acc = dadd(acc, dmul(band(y0 + 1, x0 + 1, base1, x1_in), dmul(ty, tx)));
            val = (float)acc;
        }
        // array2pil: np.array(np.clip(a, 0, 255), 'B') -- truncation; then ToDtype(scale) and `max - x` with max = 1 (white padding)
        const float cl = fminf(fmaxf(val, 0.f), 255.f);
        const unsigned v = (unsigned)cl;
        o[(size_t)Y * batch_w] = 1.0f - lut[v];
        ink = ink || v != 255u;
    }
    if (ink)                                                        // one atomic per wave: the first of its lanes that saw ink
        if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) == (int)threadIdx.x) atomicOr(flags + n, 1);
}

}  // namespace

int krk_launch_dewarp_measure(const unsigned char* crops, size_t rs, int ps, const int* desc, int n, int maxw, int maxh, const double* wts,
                              double* scratch, int* mm, int* ridge, int* centre, int* info, hipStream_t s) {
    if (n <= 0) return 0;
    if (ps != 1 && ps != 3 && ps != 4) return -4;
    const Src f{rs, ps};
    const unsigned gx = (unsigned)((maxw + 255) / 256);
    hipLaunchKernelGGL(dw_minmax_kernel, dim3(n), dim3(DW_WIDE), 0, s, crops, f, desc, mm);
    // LDS of the two Gaussian passes, sized for the tallest / widest line of the batch (r0 <= h, r1 = int(4 h + 0.5) <= w clipped)
    const int r1max = (int)(4.0 * maxh + 0.5);
    const size_t lds0 = ((size_t)(3 * maxh + 6) * G0_COLS + (size_t)maxh + 1) * sizeof(double);     // J <= h: h + 2 (h + 3) rows
    const int jpmax = (std::min(r1max, maxw) + 3) & ~3;
    const int q1 = jpmax <= 256 ? 192 : jpmax <= 512 ? 320 : 448;            // plane pitch of the instantiation: halo up to 256 / 512 / 768
    const size_t lds1 = ((size_t)4 * q1 + (size_t)jpmax + 2) * sizeof(double);
    if (lds0 > 160 * 1024 || lds1 > 160 * 1024) return -4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw_gauss0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
    const unsigned g0_threads = (unsigned)(G0_COLS * std::min(32, std::max(1, (maxh + 3) / 4)));       // 32 columns x the row blocks of the tallest line
    hipLaunchKernelGGL(dw_gauss0_kernel, dim3((unsigned)((maxw + G0_COLS - 1) / G0_COLS), n), dim3(g0_threads), lds0, s, crops, f, desc, mm, wts, scratch);
    if (jpmax > 768) return -4;
    const dim3 g1((unsigned)((maxw + G1_TILE - 1) / G1_TILE), maxh, n);
    if (q1 == 192) hipLaunchKernelGGL(dw_gauss1_kernel<192>, g1, dim3(64), lds1, s, desc, mm, wts, scratch);
    else if (q1 == 320) hipLaunchKernelGGL(dw_gauss1_kernel<320>, g1, dim3(64), lds1, s, desc, mm, wts, scratch);
    else hipLaunchKernelGGL(dw_gauss1_kernel<448>, g1, dim3(64), lds1, s, desc, mm, wts, scratch);
    hipLaunchKernelGGL(dw_unif0_kernel, dim3(gx, n), dim3(256), 0, s, desc, mm, scratch);
    const size_t ldsu = (size_t)4 * 64 * U1_PITCH * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dw_unif1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsu);
    hipLaunchKernelGGL(dw_unif1_kernel, dim3((unsigned)((maxh + 63) / 64), n), dim3(U1_THREADS), ldsu, s, desc, mm, scratch);
    hipLaunchKernelGGL(dw_ridge_kernel, dim3(gx, n), dim3(256), 0, s, desc, mm, scratch, ridge, maxw);
    hipLaunchKernelGGL(dw_centre_kernel, dim3(gx, n), dim3(256), 0, s, desc, mm, wts, ridge, centre, maxw);
    hipLaunchKernelGGL(dw_spread_kernel, dim3(n), dim3(DW_WIDE), 0, s, crops, f, desc, mm, centre, maxw, info);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int krk_launch_dewarp_apply(const unsigned char* crops, size_t rs, int ps, const int* desc, int n, int maxw, const int* mm, const int* centre,
                            const int* geo, const float* lut, int out_h, int pad, int batch_w, float* out, int* flags, hipStream_t s) {
    if (n <= 0) return 0;
    if (out_h < 1 || pad < 1 || (ps != 1 && ps != 3 && ps != 4)) return -4;
    const Src f{rs, ps};
    (void)hipMemsetAsync(flags, 0, (size_t)n * sizeof(int), s);
    hipLaunchKernelGGL(dw_apply_kernel, dim3((unsigned)((batch_w + 255) / 256), n), dim3(256), 0, s, crops, f, desc, mm, centre, maxw, geo, lut,
                       out_h, pad, batch_w, out, flags);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
