/*
 * kraken_amd.h -- C ABI of the MI355X (gfx950) line-recognition hot path.
 *
 * The reference (mittagessen/kraken) is 100 % Python and has no FFI: its
 * arithmetic is delegated to torch.nn modules.  This header is therefore the
 * boundary a kraken maintainer would bind (ctypes stub in INTEGRATION.md) to
 * replace exactly these reference call sites (paths relative to the
 * reference root):
 *
 *   krk_plan_create   <- TorchVGSLModel._parse / build_* (kraken/lib/vgsl/model.py:202-243,
 *                        570-817): the parsed layer list + state-dict tensors
 *   krk_plan_clone    <- (none) one more execution context over the same weights: kraken shares its modules between callers
 *   krk_plan_olens    <- per-layer seq_len arithmetic (kraken/lib/vgsl/layers.py:387,
 *   krk_plan_olens_w     858-859, 334); _w: with the batch's width, which Reshape.forward scales
 *                        seq_lens by (layers.py:331-332)
 *   krk_plan_out_shape <- the shape arithmetic of get_shape() on the call's tensor (layers.py:337-345 and
 *   krk_plan_out_dims     each layer's own); _dims: with the number of lines, which an Addition / Reshape
 *                        on the batch axis changes (layers.py:205-210, 313-330)
 *   krk_forward       <- MultiParamSequential.forward, i.e. `nn(x, seq_lens)`
 *                        (kraken/lib/vgsl/layers.py:44-53; model.py:488-489)
 *   krk_greedy_decode <- greedy_decoder (kraken/lib/ctc_decoder.py:35-72), optionally
 *                        fused with `(logits / T).softmax(1)`
 *                        (kraken/lib/vgsl/rpred.py:226, kraken/lib/models.py:115)
 *   krk_recognize     <- VGSLRecognitionInference._rec_predict up to the label tuples
 *                        (kraken/lib/vgsl/rpred.py:210-229)
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross the boundary;
 *   - every entry point returns KRK_OK (0) or a negative error code and never
 *     throws; krk_last_error() returns a thread-local message for the last failure;
 *   - `*_dev` pointers are HIP device pointers owned by the caller, `*_host`
 *     pointers are host memory owned by the caller;
 *   - all device work is enqueued on the caller's `stream` (a hipStream_t passed as
 *     void*); no entry point synchronises the device except where stated;
 *   - a plan is bound to one device and may be used from one stream at a time;
 *     different plans are independent.
 *
 * Semantics: masked-padding.  A batch is right-padded to a common width W; per
 * line only the first lens[n] columns are valid.  After every layer activations at
 * positions >= that line's propagated valid width are zero, GroupNorm statistics
 * cover the valid width only, and the LSTM is packed by length -- which reproduces
 * the reference's per-line (batch = 1, lens = None) result for every line of a
 * ragged batch (SURVEY.md section 8a note) and equals the reference's batched
 * result when all lines have the same width.
 */
#ifndef KRAKEN_AMD_H
#define KRAKEN_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define KRK_ABI_VERSION 3   /* 3 (round 6): + krk_plan_clone; 2 (round 6): + krk_plan_get_recurrence; round 5 had added CONVT / RESHAPE ops, krk_plan_out_dims, krk_plan_olens_w, krk_plan_set_recurrence under version 1 */

/* error codes */
#define KRK_OK            0
#define KRK_E_INVALID    -1   /* bad argument / unsupported layer configuration */
#define KRK_E_HIP        -2   /* a HIP runtime call failed */
#define KRK_E_NOMEM      -3
#define KRK_E_UNSUPPORTED -4  /* valid VGSL, but not implemented by the HIP executor */

/* layer kinds (one per reference layer wrapper in kraken/lib/vgsl/layers.py) */
#define KRK_OP_CONV       1   /* ActConv2D  layers.py:785  */
#define KRK_OP_MAXPOOL    2   /* MaxPool    layers.py:367  */
#define KRK_OP_GROUPNORM  3   /* GroupNorm  layers.py:955  */
#define KRK_OP_RESHAPE_HC 4   /* Reshape S1(1x0)1,3: fold height into channels, layers.py:285 */
#define KRK_OP_LSTM       5   /* TransposedSummarizingRNN (L{f,r,b}x) layers.py:462; act = 1: the legacy ocropy peephole cell
                               * (layers.py:72-186), w[4d+3] = its peephole vectors (i, f, o: 3 x hidden) instead of b_hh */
#define KRK_OP_LINEAR     6   /* LinSoftmax (logits, no softmax) layers.py:679 */
/* A parallel group `( a b ... )` (MultiParamParallel, layers.py:56-71; model.py:876-905) is written into the layer list as
 * PAR_BEGIN, the layers of member a, PAR_NEXT, the layers of member b, ..., PAR_END: every member reads the tensor in front of
 * PAR_BEGIN, PAR_END concatenates the members' outputs on the channel axis.  Groups nest; a serial group `[ ... ]` needs no
 * marker (it IS its layers). */
#define KRK_OP_PAR_BEGIN  7
#define KRK_OP_PAR_NEXT   8
#define KRK_OP_PAR_END    9
#define KRK_OP_ADD        10  /* Addition layers.py:188-223: cout = chunk size, kh = 0 channels / 1 height / 2 width / 3 batch */
#define KRK_OP_CONVT      11  /* ActConv2D(transposed=True) layers.py:826-834: fields as KRK_OP_CONV (sh, sw = the up-sampling
                               * factors); w[0] = the kernel of the EQUIVALENT convolution, (cout, cin, kh, kw) with both spatial
                               * axes flipped -- torch's ConvTranspose2d weight (cin, cout, kh, kw) transposed and flipped */
#define KRK_OP_RESHAPE    12  /* Reshape layers.py:285-335 (model.py:739-777), every form other than the fused RESHAPE_HC: axes in
                               * NCHW numbering (0 batch, 1 channels, 2 height, 3 width); kh = the axis that is split, kw x sh = its
                               * two parts (one of them may be -1), sw = `high`, dh = `low` (one of them == kh); cout, dw = the
                               * channels and height the layers behind it were built for (checked at the call) */

/* activations of ActConv2D (layers.py:808-825).  'sigmoid' is skipped in the
 * reference's forward (layers.py:850-852) and is therefore identical to LINEAR. */
#define KRK_ACT_LINEAR    0
#define KRK_ACT_RELU      1
#define KRK_ACT_TANH      2
#define KRK_ACT_LEAKY     3   /* torch.nn.LeakyReLU() default slope 0.01 */
#define KRK_ACT_SIGMOID   4   /* == LINEAR in forward */
#define KRK_ACT_SOFTMAX   5   /* torch.nn.Softmax(dim=1) over the channels (Cm..., the O2s... heatmap head; layers.py:814-816): f32 plan */

/* LSTM directions */
#define KRK_DIR_FWD       0
#define KRK_DIR_REV       1
#define KRK_DIR_BIDI      2

/* arithmetic modes */
#define KRK_PREC_F32      0   /* exact f32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4) */
#define KRK_PREC_BF16     1   /* OPT-IN: the KRK_PREC_BF16X3 kernels with the two cross terms dropped = plain bf16 operands, one MFMA
                               per product, f32 accumulate.  |d logit| ~1e-2: outside the 1e-3 parity gate, gated instead
                               on identical greedy strings on the fixtures; never what a precision string maps to */
#define KRK_PREC_BF16X3   2   /* split-bf16 operands (hi+lo), 3 bf16 MFMAs per product, f32 accumulate: fp32-class
                               results (|d logit| ~ 2e-5) at ~5x the f32 matrix rate; conv/projection layers only,
                               networks of the form conv(+pool) x n -> reshape -> LSTM* -> linear */

typedef struct krk_plan krk_plan;

/*
 * One VGSL layer.  Weight pointers are HOST pointers to contiguous f32 arrays in
 * the layout of the reference's state dict; they are only
 * read during krk_plan_create (packed into MFMA fragment order and uploaded).
 *
 *  CONV      cout=filters, kh,kw,sh,sw,dh,dw, act;  w[0]=co.weight (cout,cin,kh,kw), w[1]=co.bias (cout)
 *            padding is the reference's ((dh*(kh-1))/2, (dw*(kw-1))/2)   (layers.py:803)
 *  MAXPOOL   kh,kw,sh,sw (no padding, floor)
 *  GROUPNORM cout=num_groups; w[0]=layer.weight (C), w[1]=layer.bias (C); eps 1e-5
 *  RESHAPE_HC no parameters: (N,C,H,W) -> (N,H*C,1,W), feature index h*C + c
 *  LSTM      cout=hidden, direction; per direction d (0 = forward / the only one, 1 = reverse):
 *            w[4d+0]=weight_ih (4H,In)  w[4d+1]=weight_hh (4H,H)  w[4d+2]=bias_ih (4H)  w[4d+3]=bias_hh (4H)
 *            gate row order i,f,g,o (torch.nn.LSTM)
 *            kw = 0: time runs along W (Lxx); kw = 1: along H (Lxy, the reference's `transpose`, layers.py:521-523).
 *            kh = 1: summarising LSTM: only the last step of every sequence is kept (layers.py:537-539) -- with kw = 1 (L?ys) the
 *            output has height 1, with kw = 0 (L?xs) width 1 (valid widths become 1; the reference refuses seq_lens > 1
 *            there, layers.py:543-545, and so does the host side).
 *            After the height collapse (height 1, kw = 0) the layer is a plain sequence layer; on an image
 *            (height > 1, or kw = 1) every row / column is one sequence and the output is an image again
 *            (TransposedSummarizingRNN.forward on a 4-D input, layers.py:519-547) -- f32 plan only; seq_lens only with
 *            kw = 1 (columns run their full height, padding columns are zeroed afterwards), like the reference.
 *  LINEAR    cout=out features; w[0]=lin.weight (cout,In), w[1]=lin.bias (cout)
 *  PAR_BEGIN / PAR_NEXT / PAR_END  no parameters (see above).  The members' output heights and sequence/image form must
 *            agree (KRK_E_INVALID otherwise); their widths must agree for the width of the call (checked in krk_forward, as
 *            torch.cat would); valid widths behind the group are those of its LAST member (layers.py:64-66).  Exact-f32
 *            arithmetic up to the group's end (the split-bf16 kernels may take over behind it).
 *  ADD       cout = chunk, kh = axis (0 channels, 1 height, 2 width, 3 batch): out[j] = sum_k in[k*chunk + j], k < floor(size / chunk)
 *  RESHAPE   see KRK_OP_RESHAPE.  A layer that changes the batch size (ADD with kh = 3, RESHAPE touching axis 0) leaves the valid
 *            widths as they are (one per INPUT line, like the reference's seq_lens): layers behind it see full-width lines, and the
 *            ones the reference fails in with such seq_lens (packed LSTMs, a masked GroupNorm) fail at the call here too.
 */
typedef struct krk_layer {
    int op;
    int cout;
    int kh, kw, sh, sw, dh, dw;
    int act;
    int direction;
    const float* w[8];
} krk_layer;

/* Result buffers of the greedy decode, all caller-owned DEVICE memory.
 * Per line n the kernel writes counts[n] tuples into the first counts[n] slots of
 * row n of labels/starts/ends/confs (row stride = t_stride elements):
 * (label, first timestep, last timestep INCLUSIVE, max confidence in the run),
 * exactly the tuples of greedy_decoder (ctc_decoder.py:66-71). */
typedef struct krk_decode_out {
    int*   labels;    /* [N][t_stride] */
    int*   starts;    /* [N][t_stride] */
    int*   ends;      /* [N][t_stride] */
    float* confs;     /* [N][t_stride] */
    int*   counts;    /* [N]           */
    int    t_stride;  /* >= T          */
} krk_decode_out;

int         krk_abi_version(void);
const char* krk_last_error(void);

/* number of visible HIP devices (0 if none); never fails */
int krk_device_count(void);

/*
 * Builds an execution plan for `n_layers` layers applied to an input of
 * `in_channels` x `in_height` x (variable width).  Dropout/Identity layers are
 * elided by the caller.  `precision` is KRK_PREC_*.  Synchronises the device once
 * (weight upload).
 */
int krk_plan_create(const krk_layer* layers, int n_layers,
                    int in_channels, int in_height,
                    int precision, int device, krk_plan** out);
void krk_plan_destroy(krk_plan* plan);
/*
 * A second plan over the SAME packed weights (no reference analogue: the reference's modules are shared by every caller of
 * `nn`, kraken/lib/models.py:96-117; here a plan also owns per-call workspace, so concurrent batches need a plan each).  The clone
 * has its own workspace (grown on first use), events, status word, profiling and recurrence switch state; the weights are freed
 * when the last plan sharing them is destroyed, in any order.  Costs two small allocations instead of the repack + upload of
 * krk_plan_create (7 ms per plan for kraken's default recogniser: a third of an engine's start-up).
 */
int krk_plan_clone(const krk_plan* src, krk_plan** out);

/* Output geometry for an input batch of width W: channels, height, width of the
 * final layer's (N, C, H, W') output.  For a recogniser H == 1. */
int krk_plan_out_shape(const krk_plan* plan, int W, int* C, int* H, int* Wout);
/* ... for a batch of N lines, with the number of lines of the output (networks with a batch-changing ADD / RESHAPE layer; for
 * every other network Nout == N and the rest is krk_plan_out_shape's).  KRK_E_INVALID when a RESHAPE does not divide (N, W). */
int krk_plan_out_dims(const krk_plan* plan, int N, int W, int* Nout, int* C, int* H, int* Wout);

/* Host-side length propagation: olens_host[n] = valid output width of line n. */
int krk_plan_olens(const krk_plan* plan, const int* lens_host, int N, int* olens_host);
/* ... for a batch of width W: Reshape.forward scales seq_lens by (width in front) / (width behind) OF THE BATCH (layers.py:331-332),
 * so networks with a KRK_OP_RESHAPE layer need W (krk_plan_olens fails for them); identical to krk_plan_olens otherwise. */
int krk_plan_olens_w(const krk_plan* plan, const int* lens_host, int N, int W, int* olens_host);

/*
 * nn(x, lens): x_dev is (N, in_channels, in_height, W) f32 NCHW, right-padded;
 * lens_host[n] in [1, W] or NULL (all lines W wide).
 * out_dev receives the final activations:
 *   - final layer LINEAR/LSTM/RESHAPE_HC (sequence output): time-major
 *     [N][T][C] f32 (the (N,C,1,T) tensor of the reference is the permuted view);
 *   - otherwise NCHW (N, C, H', W').
 */
int krk_forward(krk_plan* plan, const float* x_dev, const int* lens_host,
                int N, int W, void* stream, float* out_dev);

/*
 * CTC best-path decode of a score tensor with element strides (sn, sc, st), i.e.
 * score(n,c,t) = scores_dev[n*sn + c*sc + t*st], N lines, C classes, T steps,
 * valid steps olens_host[n] (NULL: T).
 *   softmax == 0: confidences are the raw maxima of the input (what greedy_decoder
 *                 does with whatever it is given -- probabilities or logits);
 *   softmax == 1: the input are logits; confidences are max softmax(logits / temperature)
 *                 and ties are resolved on the probabilities like the reference.
 * If probs_dev != NULL (softmax == 1 only) the full softmax is also written there with
 * the same strides.  Argmax ties resolve to the lowest class index.
 */
int krk_greedy_decode(const float* scores_dev, long sn, long sc, long st,
                      int N, int C, int T, const int* olens_host,
                      int softmax, float temperature, float* probs_dev,
                      void* stream, const krk_decode_out* out);

/*
 * Fused nn(x, lens) -> softmax(logits / temperature) -> greedy decode for a plan
 * that ends in a LINEAR layer.  logits_dev ([N][T][C]) may be NULL if the caller
 * does not need the logits; probs_dev likewise.  olens_host (may be NULL) receives
 * the valid output widths.
 */
int krk_recognize(krk_plan* plan, const float* x_dev, const int* lens_host,
                  int N, int W, float temperature, void* stream,
                  float* logits_dev, float* probs_dev, int* olens_host,
                  const krk_decode_out* out);

/*
 * Line preprocessing on the device for rectangular crops of a fixed-height model: replaces, bit for bit, the host chain
 * the reference runs per line before the network -- im.crop(bbox) (kraken/lib/segmentation.py:1630-1643), the fixed-height
 * LANCZOS resize + white padding + ToDtype(scale) + `max - x` of ImageInputTransforms (kraken/lib/dataset/utils.py:93-152,
 * kraken/lib/functional_im_transforms.py:58-82; resampling arithmetic = Pillow's ImagingResample, 8 bits per channel).
 *   page_dev   uint8 page [page_h][page_w][channels] (channels 1 = 'L', 3 = 'RGB'), uploaded once per page
 *   boxes_dev  int32 [n][5]: x0, y0, x1, y1 of the crop (may extend past the page: zeros, like Image.crop) and the
 *              resized width out_w = int(w * out_h / h)
 *   x_dev      float (n, channels, out_h, batch_w): line k occupies columns [0, out_w_k + 2*pad), zeros to its right
 *   flags_dev  int32 [n]: 1 if the line holds any non-white pixel (the reference's flat-line rule, kraken/rpred.py:221)
 * max_in_h = tallest crop of the batch.  pad must be > 0 (the inversion `max - x` then has max = 1); KRK_E_UNSUPPORTED
 * for geometry outside the kernel's range (crop taller than 768 rows, > 96 filter taps, out_h > 128).
 */
int krk_prep_lines(const unsigned char* page_dev, int page_h, int page_w, int channels, const int* boxes_dev, int n,
                   int max_in_h, int out_h, int pad, int batch_w, float* x_dev, int* flags_dev, void* stream);
/*
 * krk_prep_lines for a page that is NOT packed [page_h][page_w][channels]: row y starts `row_stride` bytes behind row y - 1 and a
 * pixel is `pix_stride` bytes.  pix_stride 4 with channels 3 reads R, G, B of Pillow's own storage of an 'RGB' image (R, G, B, X
 * per pixel: the rows are uploaded as they lie in Pillow's memory, without the repacking of Image.tobytes / np.asarray);
 * channels 1 with pix_stride 3 or 4 gives the 1-channel model Pillow's 'L' conversion of the colour pixel,
 * (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16 -- what the reference's im.crop(box).convert('L') holds
 * (kraken/lib/dataset/utils.py:112-118; libImaging/Convert.c rgb2l).  Everything else as krk_prep_lines.
 */
int krk_prep_lines_fmt(const unsigned char* page_dev, int page_h, int page_w, long row_stride, int pix_stride, int channels,
                       const int* boxes_dev, int n, int max_in_h, int out_h, int pad, int batch_w, float* x_dev, int* flags_dev,
                       void* stream);

/*
 * The same preprocessing for line images that were cut out on the HOST (baseline / polygon extraction,
 * kraken/lib/segmentation.py:extract_polygons -- CPU geometry outside this library; or any other producer of line images):
 * the uint8 crops travel packed, 1 byte per pixel and channel, and are resized to the model height (the reference's
 * `_fixed_resize`, kraken/lib/functional_im_transforms.py:66-82), padded, scaled and inverted on the device -- replaces the
 * per-line PIL resize + float tensor of ImageInputTransforms.__call__ (kraken/lib/dataset/utils.py:93-152) without dewarping.
 *   crops_dev  uint8 images back to back, image k = [h_k][w_k][channels] at byte offset desc[k][0]
 *   desc_dev   int32 [n][4]: byte offset, w, h, resized width out_w = int(w * out_h / h)
 * Everything else as krk_prep_lines.  Bit-exact against PIL + ImageInputTransforms (tests/test_gpu_parity.py).
 */
int krk_prep_crops(const unsigned char* crops_dev, int channels, const int* desc_dev, int n, int max_in_h, int out_h, int pad,
                   int batch_w, float* x_dev, int* flags_dev, void* stream);

/*
 * CenterNormalizer dewarp on the device: what the reference runs on the host for every bounding-box line of a 1-channel model
 * (functional_im_transforms.pil_dewarp -> kraken/lib/lineest.py:26-87; arithmetic in scipy.ndimage, restated in
 * oracle/np_oracle.py:center_normalize_np and followed bit for bit, see csrc/dewarp.hip).  Two calls with a host decision between
 * them, because a line's output width depends on its measured spread:
 *   krk_dewarp_measure  centre line and spread of n packed 1-channel uint8 line images.
 *       desc_dev    int32 [n][8]: byte offset of the image in crops_dev, w, h, scratch offset (in doubles), weight-table offset (in
 *                   doubles), r0, r1, r2 = int(4 sigma + 0.5) for sigma = h/2, h, 0.3 h
 *       weights_dev fp64 Gaussian tables of the heights in the batch, per height [2 r0 + 1][2 r1 + 1][2 r2 + 1], each
 *                   exp(-x^2 / 2 sigma^2) / sum (computed by the caller with the host's exp: kraken_amd/transforms.py)
 *       scratch_dev 3 * h * w doubles per line; work_dev int32 [2 n + 2 n max_w] (min/max, ridge, centre line: kept for apply)
 *       info_dev    int32 [n][4] out: r = int(1 + 4 mad), ok (the reference's band slices are full), has ink, 0
 *       max_h <= 192 rows (r1 = 4 h <= 768 taps either side; KRK_E_UNSUPPORTED above: such lines take the reference's host transform)
 *   krk_dewarp_apply    band cut-out + bilinear scaling to out_h + uint8 truncation + white padding + / 255 + inversion.
 *       geo_dev     int32 [n][4]: r, out_w = int(out_h / (2 r) * w), use (0: the line's rows stay zero), 0
 *       x_dev       float (n, 1, out_h, batch_w); flags_dev int32 [n]: 1 if the line holds a non-white pixel
 */
int krk_dewarp_measure(const unsigned char* crops_dev, const int* desc_dev, int n, int max_w, int max_h, const double* weights_dev,
                       double* scratch_dev, int* work_dev, int* info_dev, void* stream);
int krk_dewarp_apply(const unsigned char* crops_dev, const int* desc_dev, int n, int max_w, const int* work_dev, const int* geo_dev,
                     int out_h, int pad, int batch_w, float* x_dev, int* flags_dev, void* stream);
/*
 * The same two calls for lines that are crops of ONE uploaded page instead of packed images: desc[k][0] is the byte offset of
 * the crop's first pixel in `page_dev`, a page row is `row_stride` bytes, a pixel `pix_stride` bytes (1: an 'L' page; 3 / 4: an
 * RGB / RGBX page read through Pillow's 'L' conversion, see krk_prep_lines_fmt).  No per-line packing copy on the host: the
 * lines of a bounding-box segmentation are views of the page (kraken/lib/segmentation.py:1630-1643 cuts them out one by one).
 * row_stride 0 = packed images (then pix_stride must be 1): the calls above.
 */
int krk_dewarp_measure_page(const unsigned char* page_dev, long row_stride, int pix_stride, const int* desc_dev, int n, int max_w,
                            int max_h, const double* weights_dev, double* scratch_dev, int* work_dev, int* info_dev, void* stream);
int krk_dewarp_apply_page(const unsigned char* page_dev, long row_stride, int pix_stride, const int* desc_dev, int n, int max_w,
                          const int* work_dev, const int* geo_dev, int out_h, int pad, int batch_w, float* x_dev, int* flags_dev,
                          void* stream);

/*
 * Tail of the segmenter's forward (reference kraken/lib/vgsl/spred.py:268-272): the network's class logits (C, h, w) are
 * brought to the scaled page's resolution by nearest-neighbour upsampling (F.interpolate(o, size=(H, W))) and squashed with a
 * sigmoid; one pass, (C, H, W) float32 out.
 */
int krk_upsample_sigmoid(const float* x_dev, int C, int h, int w, int H, int W, float* y_dev, void* stream);

/* Bytes of device workspace currently held by the plan (diagnostics). */
long krk_plan_workspace_bytes(const krk_plan* plan);

/*
 * Per-kernel-group timing of the last krk_forward/krk_recognize call when profiling is
 * enabled on the plan (krk_plan_set_profiling(plan, 1)).  Profiling records hipEvents
 * around each layer on the caller's stream; krk_plan_layer_ms synchronises those events.
 * Returns the number of layers written (<= cap).
 */
int krk_plan_set_profiling(krk_plan* plan, int enable);
int krk_plan_layer_ms(krk_plan* plan, float* ms_host, int cap);
/* short static name of layer i's kernel group ("conv", "lstm_xproj", ...) or NULL */
const char* krk_plan_layer_name(const krk_plan* plan, int i);
/* algorithmic FLOPs of layer i for the last call's shapes (2 * MACs over valid+padded extent) */
double krk_plan_layer_flops(const krk_plan* plan, int i);
int krk_plan_num_steps(const krk_plan* plan);

/*
 * Device-side status of the plan.  The recurrent cluster kernel (lstm_ws.hip) exchanges h_t between workgroups
 * inside a launch; every wait in it is bounded, and a wait that gives up raises a status word in mapped host memory
 * instead of hanging the device.  Call after the work of a batch has completed (stream/event synchronised):
 * KRK_OK, or KRK_E_HIP (and the word is cleared) if any batch run on this plan since the last call is invalid.
 * A krk_plan is single-stream and not thread-safe: do not use one plan from two streams or threads concurrently
 * (its workspaces are plan-owned); different plans are independent.
 */
int krk_plan_status(krk_plan* plan);
/* 1 if a forward call of this plan can launch the cluster kernel, i.e. if krk_plan_status can ever report anything: callers that
 * would synchronise only to read the status word (a plain nn(x) call) need not when this is 0. */
int krk_plan_has_exchange(const krk_plan* plan);
/* Which recurrent kernel the split-bf16 LSTM layers of THIS plan take from the next forward call on: KRK_RECURRENCE_AUTO (by
 * hidden size: the cluster kernel above 64 units) or KRK_RECURRENCE_STREAMING (lstm_x3.hip: no inter-workgroup exchange, so
 * krk_plan_status has nothing to report and krk_plan_has_exchange returns 0).  Per plan, hence per engine slot and thread-safe
 * across plans: this is how a caller re-runs ONE batch after an exchange timeout without touching other plans or the process
 * environment.  No reference analogue (torch.nn.LSTM has one CPU kernel, kraken/lib/vgsl/layers.py:513-547). */
#define KRK_RECURRENCE_AUTO 0
#define KRK_RECURRENCE_STREAMING 1
int krk_plan_set_recurrence(krk_plan* plan, int variant);
/* The plan's current setting (KRK_RECURRENCE_*; negative = error): lets a caller that switches a plan for one retry put back
 * what was there instead of assuming KRK_RECURRENCE_AUTO. */
int krk_plan_get_recurrence(const krk_plan* plan);

/* Cross-batch scheduling for callers that keep several plans in flight on separate streams
 * (kraken_amd/engine.py; no reference analogue -- the reference runs one batch at a time,
 * kraken/lib/vgsl/rpred.py:210-229).  `krk_plan_front_event` returns a hipEvent_t (as void*, owned by
 * the plan) that every krk_forward / krk_recognize call records once the batch's convolution block and
 * first LSTM input projection have been enqueued; `krk_plan_wait_front(plan, ev)` makes the NEXT call on
 * `plan` wait for `ev` before its first kernel (one shot).  Chaining batch k+1 behind batch k's front event
 * keeps the full-chip convolution blocks of different batches from time-slicing each other while the
 * 32-CU recurrent kernels of batch k overlap the convolutions of batch k+1. */
void* krk_plan_front_event(krk_plan* plan);
int krk_plan_wait_front(krk_plan* plan, void* event);

#ifdef __cplusplus
}
#endif
#endif /* KRAKEN_AMD_H */
