#!/usr/bin/env python
"""
Generates the golden fixtures in this directory by running the UNMODIFIED reference
(mittagessen/kraken mounted at /root/reference) on CPU, fp32.  Run in the authoring container:

    python tests/golden/make_golden.py

Nothing here runs on the GPU box; the fixtures travel instead.  What is pinned:

  overfit.npz      the reference's own known-answer test (tests/test_rpred.py:352-358, :453-462):
                   weights of tests/resources/overfit.mlmodel (read with kraken_amd.io, validated by
                   reproducing the test's exact strings through the reference's TorchVGSLModel +
                   TorchSeqRecognizer + rpred), the preprocessed line tensor, logits, softmax
                   outputs, greedy_decoder tuples, codec output and the expected strings.
  overfit_models.npz  overfit_newpoly.mlmodel / overfit_bl{,_newpoly}.safetensors of the reference's test resources: the
                   fixture line through both transform branches -> logits, softmax, greedy tuples, strings.
  bench_a.npz      BENCH-A, torch.manual_seed(0) + TorchVGSLModel init: state-dict digests,
  bench_b.npz      input digests, logits of selected lines, per-step argmax / max-prob and
                   greedy_decoder tuples for equal-width batches, and per-line (batch = 1) results for
                   the ragged widths {401, 613, 800} -- the parity target for masked padding.
  layers.npz       single-layer networks through the reference's layer wrappers: odd/even kernels,
  image_lstm.npz   LSTMs over image rows / columns (Lxx, Lxy on 4-D inputs) and a scaled-down BLLA segmenter.
                   strides, dilation (incl. the Cr4,2,*,4,2 form of tests/test_vgsl.py:71), max-pool
                   floor cases, masked GroupNorm, the S1(1x0)1,3 reshape, f/r/b LSTMs with ragged lens.
  bench_lines.npz  BASELINE configs 2 and 4 line by line: kraken's tuples, strings and top-2 margins for all 256 lines of
                   the benchmark tensor and for 1024 distinct ragged lines (batch 1 each).
  bench_lines_r6.npz  BASELINE config 3's rank shard line by line (2048 DISTINCT lines 1x48x1200) and 64 lines 1x120x1200 through
                   kraken's default height-120 spec (kraken/configs/vgsl.py:102): tuples, strings, margins (+ logits of 4 lines).
  big_lstm.npz     recurrent layers above 768 hidden units (1024 bidirectional, 1280 forward, 832 peephole), ragged lengths.
  codec.npz        PytorchCodec.decode / encode known answers incl. multi-label codes.
  transforms.npz   ImageInputTransforms outputs (dewarp + fixed-height paths) for synthetic line images.
"""
import hashlib
import json
from collections import defaultdict
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refshim  # noqa: E402

_refshim.install()

from kraken.lib import vgsl as ref_vgsl  # noqa: E402
from kraken.lib.codec import PytorchCodec as RefCodec  # noqa: E402
from kraken.lib.ctc_decoder import greedy_decoder as ref_greedy  # noqa: E402
from kraken.lib.models import TorchSeqRecognizer as RefRecognizer  # noqa: E402

from tests.specs import BENCH_A, BENCH_B, bench_codec  # noqa: E402
from kraken_amd.specs import DEFAULT_H120  # noqa: E402

RES = os.path.join(_refshim.REFERENCE_ROOT, 'tests', 'resources')


def digest(t) -> str:
    a = np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t)
    return hashlib.sha256(a.tobytes()).hexdigest()


def tuples_to_arr(dec):
    """list of lists of (label,start,end,conf) -> (flat float64 [n,4], counts)"""
    flat = [list(t) for line in dec for t in line]
    return np.array(flat, dtype=np.float64).reshape(-1, 4), np.array([len(line) for line in dec], dtype=np.int32)


def synth_input(n, w, seed=1234, h=48, c=1):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, c, h, w, generator=g)


@torch.inference_mode()
def bench_fixture(spec, path, seed=0, cases=None):
    torch.manual_seed(seed)
    net = ref_vgsl.TorchVGSLModel(vgsl=spec, codec=bench_codec())
    net.eval()
    out = {'spec': spec, 'seed': seed,
           'state_digest': json.dumps({k: digest(v) for k, v in net.state_dict().items()})}
    rec = RefRecognizer(net, device='cpu')
    # equal-width batches
    for tag, n, w, keep in cases or (('n4w400', 4, 400, [0, 3]), ('n16w800', 16, 800, [0, 7]), ('n4w1200', 4, 1200, [1])):
        x = synth_input(n, w)
        lens = torch.tensor([w] * n)
        logits, olens = net.nn(x, lens)
        probs = logits.softmax(1).squeeze(2)
        conf, lab = probs.max(dim=1)
        dec = ref_greedy(probs, olens)
        flat, counts = tuples_to_arr(dec)
        out[f'{tag}_xdigest'] = digest(x)
        out[f'{tag}_olens'] = olens.numpy().astype(np.int32)
        out[f'{tag}_keep'] = np.array(keep)
        out[f'{tag}_logits'] = logits[keep].squeeze(2).numpy()
        out[f'{tag}_labels'] = lab.numpy().astype(np.int32)
        out[f'{tag}_conf'] = conf.numpy()
        out[f'{tag}_tuples'] = flat
        out[f'{tag}_counts'] = counts
        out[f'{tag}_logit_abs_sum'] = logits.abs().sum(dim=(1, 2, 3)).numpy()
        strings = rec.predict_string(x, lens)
        out[f'{tag}_strings'] = json.dumps(strings)
    # ragged widths, each line on its own (batch = 1, lens = None): the reference's per-line result
    widths = [800, 613, 401]
    xr = synth_input(len(widths), 800, seed=4321)
    out['ragged_widths'] = np.array(widths, dtype=np.int32)
    out['ragged_xdigest'] = digest(xr)
    for i, w in enumerate(widths):
        xi = xr[i:i + 1, :, :, :w].contiguous()
        logits, _ = net.nn(xi)
        probs = logits.softmax(1).squeeze(2)
        flat, counts = tuples_to_arr(ref_greedy(probs))
        out[f'ragged{i}_logits'] = logits.squeeze(2)[0].numpy()
        out[f'ragged{i}_tuples'] = flat
        out[f'ragged{i}_counts'] = counts
    # the same ragged lines through the reference's BATCHED path (padding not masked): documents
    # that the reference itself is not batch-invariant (SURVEY.md 8a note)
    xb = xr.clone()
    for i, w in enumerate(widths):
        xb[i, ..., w:] = 0
    lb, ob = net.nn(xb, torch.tensor(widths))
    out['ragged_batched_olens'] = ob.numpy().astype(np.int32)
    out['ragged_batched_maxdiff'] = np.array([
        float((lb[i, :, 0, :ob[i]] - torch.from_numpy(out[f'ragged{i}_logits'])[:, :ob[i]]).abs().max())
        for i in range(len(widths))])
    np.savez_compressed(path, **out)
    print('wrote', path, {k: (v.shape if hasattr(v, 'shape') else '') for k, v in out.items() if 'logits' in k})


@torch.inference_mode()
def bench_lines_fixture(path, seed=0):
    """
    BASELINE.json configs 2 and 4 pinned LINE BY LINE against kraken: the reference's greedy tuples,
    its predict_string output and the smallest top-2 logit margin of every line, for
      cfg2  all 256 lines of synth_input(256, 1200) -- the tensor bench.py times (kraken/lib/models.py:138-149);
      cfg4  1024 DISTINCT ragged lines, W ~ U{400..2400} (RandomState(40), sorted as the width-bucketing pipeline sees them),
            line i = synth_input(1, W_i, seed=50000 + i), each through the reference at batch 1 (its per-line rpred result,
            kraken/lib/vgsl/rpred.py:129-131).
    """
    torch.manual_seed(seed)
    net = ref_vgsl.TorchVGSLModel(vgsl=BENCH_A, codec=bench_codec())
    net.eval()
    rec = RefRecognizer(net, device='cpu')
    out = {'spec': BENCH_A, 'seed': seed,
           'state_digest': json.dumps({k: digest(v) for k, v in net.state_dict().items()})}

    def one(x, lens):
        logits, olens = net.nn(x, lens)
        probs = logits.softmax(1).squeeze(2)
        dec = ref_greedy(probs, olens)
        top2 = logits.squeeze(2).topk(2, dim=1).values            # [n, 2, T]
        margin = (top2[:, 0] - top2[:, 1])
        if olens is not None:
            for i, l in enumerate(olens.tolist()):
                margin[i, l:] = float('inf')
        return dec, margin.min(dim=1).values, rec.predict_string(x, lens)

    x = synth_input(256, 1200)
    dec, mar, strs = [], [], []
    for lo in range(0, 256, 16):
        d, m, s = one(x[lo:lo + 16], torch.tensor([1200] * 16))
        dec += d
        mar.append(m)
        strs += s
    flat, counts = tuples_to_arr(dec)
    out['cfg2_xdigest'] = digest(x)
    out['cfg2_tuples'] = flat.astype(np.float32)
    out['cfg2_counts'] = counts
    out['cfg2_margin'] = torch.cat(mar).numpy()
    out['cfg2_strings'] = json.dumps(strs)

    rng = np.random.RandomState(40)
    widths = np.sort(rng.randint(400, 2401, size=1024))
    dec, mar, strs, dig = [], [], [], hashlib.sha256()
    for i, w in enumerate(widths.tolist()):
        xi = synth_input(1, w, seed=50000 + i)
        dig.update(np.ascontiguousarray(xi.numpy()).tobytes())
        d, m, s = one(xi, None)
        dec += d
        mar.append(m)
        strs += s
    flat, counts = tuples_to_arr(dec)
    out['cfg4_widths'] = widths.astype(np.int32)
    out['cfg4_xdigest'] = dig.hexdigest()
    out['cfg4_tuples'] = flat.astype(np.float32)
    out['cfg4_counts'] = counts
    out['cfg4_margin'] = torch.cat(mar).numpy()
    out['cfg4_strings'] = json.dumps(strs)
    np.savez_compressed(path, **out)
    print('wrote', path, 'cfg2 tuples', out['cfg2_tuples'].shape, 'cfg4 tuples', out['cfg4_tuples'].shape,
          'min margins', float(out['cfg2_margin'].min()), float(out['cfg4_margin'].min()))


IMAGE_LSTM_CASES = {
    # LSTMs over image rows (x) and columns (y) -- TransposedSummarizingRNN on 4-D inputs -- and a scaled-down
    # BLLA segmenter (reference default spec kraken/configs/vgsl.py:122 + the O2l heatmap head, model.py:806-811)
    'lstm_x_img':  ('[1,5,0,3 Lbx6]', 2, 9, None),
    'lstm_y_img':  ('[1,5,0,3 Lby6]', 2, 9, None),
    'lstm_fy_img': ('[1,7,0,4 Lfy5]', 1, 6, None),
    'lstm_xy_h1':  ('[1,1,0,6 Lby4]', 2, 5, None),
    'lstm_fys':    ('[1,6,0,3 Lfys5]', 2, 8, None),
    'lstm_bys':    ('[1,5,0,2 Lbys4]', 2, 7, None),
    'tess_style':  ('[1,0,0,1 Cr3,3,8 Mp3,3 Lfys16 Lbx8 O1c5]', 3, 47, [47, 30, 12]),
    'lby_ragged':  ('[1,10,0,1 Cr3,3,8 Lby4 Cr3,3,8]', 2, 21, [21, 13]),
    'blla_small':  ('[1,48,0,3 Cr7,7,16,2,2 Gn8 Cr3,3,32,2,2 Gn8 Cr3,3,32 Gn8 Lbx8 Lby8 Cr1,1,8 Gn8 Lby8 Lbx8 O2l3]', 1, 70, None),
    'blla_batch2': ('[1,20,0,3 Cr7,7,8,2,2 Gn4 Lbx4 Lby4 Cr1,1,8 Gn4 Lby4 Lbx4 O2l2]', 2, 33, None),
}


X3_NETWORK_CASES = {
    # whole small networks for the split-bf16 kernels, outputs made by the REFERENCE's modules: the tap-as-K convolutions
    # (first layer; five-group packing at kw 11 / 12 / 13 = window shifts 2 and 3; six-group at kw 15), channel counts that are
    # not powers of two, fused pools, a GroupNorm consumer, and stacks of recurrent layers (tile-time-major rows between them)
    'x3_taps13':    ('[1,20,0,1 Cr3,13,8 Cr3,13,12 Mp2,2 Cr3,5,32 S1(1x0)1,3 Lfx40 Lrx24 Lbx16 O1c21]', 3, 150, [150, 97, 40]),
    'x3_taps11':    ('[1,16,0,1 Cr3,11,20 Mp2,2 Cr3,11,28 Cr3,3,16 S1(1x0)1,3 Lbx24 Lbx40 O1c40]', 3, 261, [261, 260, 131]),
    'x3_taps12_gn': ('[1,12,0,1 Cr3,12,16 Cr3,12,32 Gn8 Mp2,2 Cr3,3,16 S1(1x0)1,3 Lbx16 Lfx32 O1c8]', 2, 133, [133, 70]),
    'x3_taps15':    ('[1,8,0,1 Ct1,3,4 Cs3,15,16 Cl3,3,16 S1(1x0)1,3 Lfx16 O1c5]', 2, 96, None),
    'x3_c32_taps':  ('[1,24,0,1 Cr3,13,32 Mp2,2 Cr3,13,32 Mp2,2 Cr3,9,32 S1(1x0)1,3 Lbx56 Lbx24 O1c50]', 2, 400, [400, 233]),
    # round 5: colour lines (3 input channels) through the first-layer kernel: pooled + tap kernel behind it (the RGB recogniser's
    # shape), an unpooled first layer, kernel height 1, fewer than 32 filters, a first layer whose consumer is not the tap kernel
    'x3_rgb_pool':  ('[1,16,0,3 Cr3,13,32 Mp2,2 Cr3,13,32 Mp2,2 Cr3,9,32 S1(1x0)1,3 Lbx40 O1c30]', 3, 300, [300, 171, 64]),
    'x3_rgb_flat':  ('[1,12,0,3 Ct3,11,28 Cr3,11,32 Mp2,2 Cr3,3,16 S1(1x0)1,3 Lfx24 O1c9]', 2, 150, [150, 77]),
    'x3_rgb_kh1':   ('[1,8,0,3 Cl1,9,16 Cr3,3,16 S1(1x0)1,3 Lbx16 O1c8]', 2, 131, None),
    'x3_rgb_strid': ('[1,16,0,3 Ct3,3,16 Cr3,7,48,1,2 Cl1,1,32 S1(1x0)1,3 Lbx8 O1c7]', 2, 97, [97, 50]),
}


BREADTH_CASES = {
    # round 4: VGSL forms the executor learned after the first fixtures were made
    # 'G' cell: the reference parses it and builds the same torch.nn.LSTM (layers.py:504-511); layer name G_<idx>
    'g_alias':       ('[1,1,0,12 Gbx10]', 3, 17, [17, 9, 4]),
    'g_stack':       ('[1,4,0,2 Cr3,3,4 S1(1x0)1,3 Gfx8 Lbx6 O1c5]', 3, 23, [23, 15, 8]),
    # hidden sizes above 256: the generic-width recurrent kernel (lstm_big_kernel), also between split-bf16 layers
    'lstm_b_h300':   ('[1,1,0,24 Lbx300]', 3, 19, [19, 11, 5]),
    'lstm_f_h520':   ('[1,1,0,16 Lfx520 O1c9]', 2, 13, None),
    'big_stack':     ('[1,8,0,1 Cr3,3,8 Cr3,3,16 S1(1x0)1,3 Lbx264 Lbx24 O1c12]', 3, 40, [40, 27, 9]),
}


SIZES_R6_CASES = {
    # round 6: sizes the split-bf16 plan had refused or sent to the exact-f32 kernels.  Hidden sizes that are not a multiple of 8 (every
    # direction is written Hp units wide; above 64 units the cluster kernel), feature counts that are not a multiple of 8 or 16, a
    # convolution stack without 16-channel K blocks in front of recurrent layers (f32 convolutions, split-bf16 sequence part)
    'odd_h70':     ('[1,2,0,1 Cr3,3,8 Cr3,3,16 S1(1x0)1,3 Lbx70 Lbx70 O1c12]', 3, 40, [40, 27, 9]),
    'odd_h75_99':  ('[1,2,0,1 Cr3,3,8 Cr3,3,16 S1(1x0)1,3 Lfx75 Lrx99 O1c9]', 3, 33, [33, 20, 7]),
    'odd_tiny':    ('[1,2,0,1 Cr3,3,8 Cr3,3,16 S1(1x0)1,3 Lbx6 Lfx5 Lrx3 O1c5]', 2, 25, None),
    'feat36':      ('[1,6,0,1 Cr3,5,16 Mp2,2 Cr3,3,16 Cr3,3,12 S1(1x0)1,3 Lbx20 O1c9]', 3, 44, [44, 30, 21]),
    'ch24':        ('[1,16,0,1 Cr3,3,24 Mp2,2 Cr3,3,48 Cr3,3,40 S1(1x0)1,3 Lbx20 Lfx12 O1c9]', 3, 50, [50, 33, 17]),
}


FORMS_R5_CASES = {
    # round 5: the VGSL forms that were still refused.  ocropy's peephole cell (layers.py:72-186, 'o'): always bidirectional, no
    # biases, a constant 1 in front of the input; ragged batches against the reference's per-line result (it has no batched form)
    'peep_b':      ('[1,1,0,12 Lbxo10]', 3, 17, [17, 9, 4]),
    'peep_stack':  ('[1,8,0,1 Cr3,3,4 S1(1x0)1,3 Lbxo8 Lbx6 O1c5]', 3, 23, [23, 15, 8]),
    'peep_y':      ('[1,6,0,2 Lbyo4]', 2, 9, None),
    'peep_h40':    ('[1,1,0,20 Lbxo40 O1c7]', 2, 30, None),
    # transposed convolutions (ActConv2D(transposed=True), layers.py:826-834; model.py:701-712): up-sampling strides, an even kernel
    # (padding (k - 1) // 2 is one short of "same"), dilation, and inside a recogniser
    'ct_up2':      ('[1,10,0,2 CTr3,3,6,2,2]', 2, 19, [19, 11]),
    'ct_even':     ('[1,8,0,3 CTl4,2,5,2,3]', 2, 13, None),
    'ct_dil':      ('[1,9,0,2 CTt3,3,4,1,2,2,1]', 1, 15, None),
    'ct_net':      ('[1,12,0,1 Cr3,3,8 Mp2,2 CTr3,3,8,2,2 Cr3,3,4 S1(1x0)1,3 Lbx8 O1c5]', 3, 30, [30, 21, 8]),
    # Addition over the width (layers.py:188-223): 17 columns in pieces of 5, the remainder dropped; seq_lens do not survive it in the
    # reference either (they are handed through unchanged and exceed the new width)
    'add_w':       ('[1,6,0,3 Cr3,3,4 A2,5]', 2, 17, None),
    'add_w_seq':   ('[1,1,0,6 A2,8 Lfx5 O1c4]', 2, 27, None),
    # Reshape in general (layers.py:285-335; build_reshape, model.py:739-777): an axis split in two, one part rotated in front of
    # another axis and merged with it.  Fifth entry: seq_lens whose values behind the network are recorded from the reference's batched
    # call (Reshape scales them by the ratio of the BATCH's widths; the tensors are compared without seq_lens)
    'rs_c_b2h':    ('[1,6,0,4 S3(2x2)3,1]', 2, 7, None, [7, 5]),              # channels (2 x 2): the minor part in front of the height
    'rs_c_a2h':    ('[1,6,0,4 S3(2x2)1,3]', 2, 7, None, [7, 5]),              # ... the major part
    'rs_h_b2w':    ('[1,6,0,4 S1(2x3)1,2]', 2, 7, None, [7, 5]),              # height (2 x 3) into the width: three times as wide
    'rs_h_a2w':    ('[1,6,0,4 S1(2x3)2,1]', 2, 7, None, [7, 5]),
    'rs_w_b2c':    ('[1,6,12,4 S2(3x4)2,3]', 2, 12, None, [12, 10]),          # a fixed width (3 x 4) into the channels
    'rs_w_a2h':    ('[1,6,12,4 S2(0x4)1,2]', 2, 12, None, [12, 10]),          # ... with one part left to the tensor (-1)
    'rs_same':     ('[1,6,0,4 S1(2x3)1,1]', 2, 7, None, [7, 5]),              # high == low: the two parts change places
    'rs_alt_hc':   ('[1,4,0,2 Cr3,3,3 S1(4x1)3,1 Lbx5 O1c4]', 2, 11, None, [11, 9]),   # the height collapse spelled the other way round
    'rs_seq':      ('[1,1,0,6 Lbx4 S3(2x4)3,1 Cr3,3,5]', 2, 9, None, [9, 7]),          # behind a sequence layer: an image again
    'rs_lin_img':  ('[1,8,0,2 Cr3,3,8 S3(2x4)1,3 O1c5]', 2, 11, None, [11, 6]),      # a linear layer over an image of 16 rows (layers.py:710-722)
    'rs_net':      ('[1,8,0,1 Cr3,3,4 S1(2x4)1,2 Cr3,3,6 Mp2,2 S1(1x0)1,3 Lbx6 O1c5]', 3, 14, None, None),
    # ... on the batch axis, and Addition over the batch (layers.py:188-223): the number of lines changes, the seq_lens keep
    # counting the input's lines
    'rs_n_b2c':    ('[4,6,0,2 S0(2x2)0,3]', 4, 9, None, [9, 7, 6, 5]),
    'rs_h2n':      ('[1,6,0,2 S1(2x3)0,1]', 3, 9, None, [9, 7, 6]),
    'rs_n_net':    ('[4,6,0,2 S0(2x2)0,3 Cr3,3,4 Mp2,2]', 4, 9, None, [9, 7, 6, 5]),
    'add_n':       ('[1,6,0,2 Cr3,3,3 A0,2]', 5, 9, None, [9, 7, 6, 5, 4]),
    'add_n_seq':   ('[1,1,0,6 Lfx5 A0,2 O1c4]', 4, 9, None, [9, 7, 6, 5]),
    'add_n_gn':    ('[1,6,0,4 A0,3 Gn2 Cr3,3,2]', 6, 9, None, None),
}

GROUP_CASES = {
    # round 4: nested serial `[ ... ]` and parallel `( ... )` groups (model.py:847-905, layers.py:56-71), Addition (layers.py:188-223),
    # x-axis summarising LSTMs (layers.py:537-545).  The first spec is the reference's own (tests/test_vgsl.py:71)
    'par_nested':   ('[1,48,0,1 Cr4,2,1,4,2 ([Cr4,2,1,1,1 Do Cr3,3,2,1,1] [Cr4,2,1,1,1 Cr3,3,2,1,1 Do]) S1(1x0)1,3 Lbx2 Do0.5 Lbx2]',
                     2, 40, [40, 25]),
    'par_bare':     ('[1,12,0,1 (Cr3,3,4 Cr5,5,6 Ct1,1,2) Mp2,2 S1(1x0)1,3 Lbx8 O1c5]', 3, 30, [30, 21, 8]),
    'par_residual': ('[1,10,0,2 Cr3,3,8 (I [Cr3,3,8 Cl3,3,8]) A3,8 Cr3,3,4]', 2, 19, None),
    'par_seq':      ('[1,1,0,12 (Lfx6 Lbx5) O1c7]', 3, 17, [17, 9, 4]),
    'par_in_par':   ('[1,8,0,1 ([Cr3,3,4 (Cr3,3,2 Cr1,1,3)] Cr3,3,4) Gn3 S1(1x0)1,3 Lbx6]', 2, 21, None),
    'par_then_x3':  ('[1,16,0,1 (Cr3,3,8 Cr5,5,8) Cr3,3,16 Mp2,2 Cr3,3,16 S1(1x0)1,3 Lbx16 Lbx8 O1c10]', 3, 44, [44, 30, 21]),
    'par_pooled':   ('[1,12,0,1 ([Cr3,3,4 Mp2,2] [Mp2,2 Cr3,3,4]) S1(1x0)1,3 Lfx6]', 3, 33, [33, 20, 9]),
    'add_h':        ('[1,12,0,2 Cr3,3,4 A1,4 S1(1x0)1,3 Lfx5]', 2, 15, None),
    'add_c_rem':    ('[1,6,0,7 A3,3 Cr3,3,2]', 2, 11, None),
    'add_seq':      ('[1,1,0,12 Lbx6 A3,6 O1c4]', 3, 17, [17, 9, 4]),
    'sum_x_img':    ('[1,6,0,2 Lfxs4]', 2, 13, None),
    'sum_x_seq':    ('[1,1,0,9 Lbxs5 O1c3]', 3, 11, None),
    # channel-softmax convolutions (nl 'm', layers.py:814-816) and the softmax heatmap head O2s (model.py:806-811)
    'conv_softmax': ('[1,8,0,2 Cm3,3,5 Cr3,3,4]', 2, 19, [19, 12]),
    'heat_softmax': ('[1,12,0,3 Cr3,3,8 Mp2,2 O2s4]', 2, 21, None),
    # the clstm legacy layout: a constant 1 in front of every input vector, no biases (layers.py:498-511, 522-524)
    'clstm':        ('[1,1,0,12 Lbxc10]', 3, 17, [17, 9, 4]),
    'clstm_stack':  ('[1,8,0,1 Cr3,3,4 S1(1x0)1,3 Lfxc8 Lbxc6 O1c5]', 3, 23, [23, 15, 8]),
    'clstm_y':      ('[1,6,0,2 Lfyc4]', 2, 9, None),
}


@torch.inference_mode()
def bench_lines_r6_fixture(path, seed=0, only=None):
    """
    VERDICT r5 item 7.  cfg3: one rank's shard of BASELINE config 3 -- 2048 DISTINCT lines 1x48x1200, line i = row i % 16 of
    synth_input(16, 1200, seed=30000 + i // 16) -- through the unmodified reference in batches of 16 (equal widths: the batched result
    is the per-line result).  h120: 64 lines 1x120x1200 (synth_input(64, 1200, seed=1200, h=120)) through kraken's DEFAULT
    recognition spec (kraken/configs/vgsl.py:102, height 120) with a 256-class output layer and portable weights; logits of its first
    4 lines.  R6_ONLY=h120 (or cfg3) recomputes one of the two and keeps the other from the existing file.
    """
    from tests.helpers import portable_weights
    out = {}
    if only and os.path.exists(path):
        out = {k: v for k, v in np.load(path, allow_pickle=False).items()}
    for tag, spec, nlines, h in (('cfg3', BENCH_A, 2048, 48), ('h120', DEFAULT_H120, 64, 120)):
        if only and tag not in only:
            continue
        torch.manual_seed(seed)
        net = ref_vgsl.TorchVGSLModel(vgsl=spec, codec=bench_codec())
        if tag == 'h120':
            # kraken's orthogonal LSTM init is a LAPACK QR: for the 800 x 960 input weights of this spec its bits differ between
            # machines, so this case draws its weights element-wise (tests/helpers.py: portable_weights)
            portable_weights(net, seed=120)
        net.eval()
        rec = RefRecognizer(net, device='cpu')
        out[f'{tag}_spec'] = spec
        out[f'{tag}_state_digest'] = json.dumps({k: digest(v) for k, v in net.state_dict().items()})
        dec, mar, strs, dig = [], [], [], hashlib.sha256()
        for lo in range(0, nlines, 16):
            x = synth_input(16, 1200, seed=30000 + lo // 16, h=h) if tag == 'cfg3' else synth_input(64, 1200, seed=1200, h=h)[lo:lo + 16]
            dig.update(np.ascontiguousarray(x.numpy()).tobytes())
            lens = torch.tensor([1200] * 16)
            logits, olens = net.nn(x, lens)
            dec += ref_greedy(logits.softmax(1).squeeze(2), olens)
            top2 = logits.squeeze(2).topk(2, dim=1).values
            mar.append((top2[:, 0] - top2[:, 1]).min(dim=1).values)
            strs += rec.predict_string(x, lens)
            if tag == 'h120' and lo == 0:
                out['h120_logits4'] = logits[:4].numpy().astype(np.float32)
        flat, counts = tuples_to_arr(dec)
        out[f'{tag}_xdigest'] = dig.hexdigest()
        out[f'{tag}_tuples'] = flat.astype(np.float32)
        out[f'{tag}_counts'] = counts
        out[f'{tag}_margin'] = torch.cat(mar).numpy()
        out[f'{tag}_strings'] = json.dumps(strs)
        print(tag, 'tuples', out[f'{tag}_tuples'].shape, 'min margin', float(out[f'{tag}_margin'].min()), flush=True)
    np.savez_compressed(path, **out)
    print('wrote', path)


@torch.inference_mode()
def big_lstm_fixture(path):
    """
    Hidden sizes above 768 (the reference builds any size, model.py:570-597): three one-layer recognisers through
    the unmodified reference with ragged seq_lens -- 1024 bidirectional (cell state in HBM), 1280 forward (h in HBM too), 832 with
    the ocropy peephole cell.  The weights are `portable_weights` (tests/helpers.py: 33 MB per case otherwise); inputs and logits.
    """
    from tests.helpers import portable_weights
    out = {}
    # 'classic': kraken's classic recognition spec (48 rows, two pools, the height collapse written out as S1(1x12)1,3): since round 6
    # that spelling takes the fused collapse like S1(1x0)1,3 (kraken_amd/vgsl.py)
    cases = {'bidi1024': '[1,1,0,16 Lbx1024 O1c8]', 'fwd1280': '[1,1,0,16 Lfx1280 O1c8]', 'peep832': '[1,1,0,16 Lbxo832 O1c8]',
             'classic': '[1,48,0,1 Cr3,3,32 Do0.1,2 Mp2,2 Cr3,3,64 Do0.1,2 Mp2,2 S1(1x12)1,3 Lbx100 Do O1c50]'}
    for k, (tag, spec) in enumerate(cases.items()):
        net = ref_vgsl.TorchVGSLModel(vgsl=spec)
        portable_weights(net, seed=900 + k)
        net.eval()
        x = synth_input(5, 23, seed=7000 + k, h=1, c=16) if tag != 'classic' else synth_input(5, 92, seed=7000 + k, h=48, c=1)
        # (the reference's peephole cell does not take packed sequences, layers.py:176: that case runs full-width lines, no seq_lens)
        lens = torch.tensor([23, 9, 17, 1, 23] if 'peep' not in tag else [23] * 5) * (4 if tag == 'classic' else 1)
        for i, l in enumerate(lens.tolist()):
            x[i, ..., l:] = 0
        y, olens = net.nn(x, lens if 'peep' not in tag else None)
        olens = lens if olens is None else olens
        if tag == 'classic':
            # a convolutional stack: the reference's BATCHED result with padding depends on the batch (the padding bleeds through
            # bias + ReLU: DESIGN.md section 1); the parity target is its per-line result, line by line at batch 1
            for i, l in enumerate(lens.tolist()):
                yi, _ = net.nn(x[i:i + 1, ..., :l].contiguous())
                y[i] = 0
                y[i, ..., :yi.shape[3]] = yi[0]
        out[f'{tag}_spec'] = spec
        out[f'{tag}_seed'] = 900 + k
        out[f'{tag}_x'] = x.numpy()
        out[f'{tag}_lens'] = lens.numpy()
        out[f'{tag}_y'] = y.numpy()
        out[f'{tag}_olens'] = olens.numpy()
        out[f'{tag}_state_digest'] = json.dumps({n: digest(v) for n, v in net.state_dict().items()})
        print(tag, tuple(y.shape), float(y.abs().max()))
    np.savez_compressed(path, **out)
    print('wrote', path)


@torch.inference_mode()
def layer_fixture(path, cases=None):
    cases = cases or {
        'conv_odd':      ('[1,9,0,3 Cr3,5,7]', 2, 37, None),
        'conv_even_str': ('[1,12,0,1 Cr4,2,5,4,2]', 2, 41, None),
        'conv_stride2':  ('[1,30,0,1 Cr3,3,8,2,2]', 2, 45, [45, 31]),
        'conv_dilated':  ('[1,10,0,2 Ct3,3,6,1,1,2,2]', 2, 33, None),
        'conv_leaky':    ('[1,8,0,2 Clr3,3,4]', 1, 19, None),
        'conv_linear':   ('[1,8,0,2 Cl1,1,40]', 1, 19, None),
        'conv_sigmoid':  ('[1,8,0,2 Cs3,3,4]', 1, 19, None),
        'conv_wide':     ('[1,6,0,4 Cr3,13,33]', 3, 70, [70, 53, 9]),
        'pool_floor':    ('[1,9,0,2 Mp2,2]', 2, 37, [37, 20]),
        'pool_3x2_s2x3': ('[1,11,0,2 Mp3,2,2,3]', 2, 40, None),
        'conv_pool':     ('[1,16,0,1 Cr3,3,8 Mp2,2]', 3, 50, [50, 33, 17]),
        'gn_full':       ('[1,6,0,8 Gn4]', 2, 21, None),
        'gn_masked':     ('[1,6,0,8 Gn4]', 3, 21, [21, 13, 5]),
        'conv_gn_pool':  ('[1,12,0,1 Cr3,3,16 Gn8 Mp2,2]', 3, 44, [44, 30, 21]),
        'reshape':       ('[1,5,0,3 S1(1x0)1,3]', 2, 7, None),
        'lstm_f':        ('[1,1,0,12 Lfx10]', 3, 17, [17, 9, 4]),
        'lstm_r':        ('[1,1,0,12 Lrx10]', 3, 17, [17, 9, 4]),
        'lstm_b':        ('[1,1,0,12 Lbx10]', 3, 17, [17, 9, 4]),
        'lstm_b_h13':    ('[1,1,0,7 Lbx13]', 2, 11, None),
        'lstm_stack':    ('[1,4,0,2 Cr3,3,4 S1(1x0)1,3 Lbx8 Lfx6 O1c5]', 3, 23, [23, 15, 8]),
        'linear':        ('[1,1,0,20 O1c9]', 2, 13, None),
        'two_linear':    ('[1,1,0,20 O1c16 O1c36]', 2, 13, None),
    }
    out = {'cases': json.dumps({k: {'spec': v[0], 'n': v[1], 'w': v[2], 'lens': v[3], **({'lens_probe': v[4]} if len(v) > 4 and v[4] else {})}
                                for k, v in cases.items()})}
    for name, (spec, n, w, lens, *probe) in cases.items():
        torch.manual_seed(hash(name) % 1000 if False else sum(map(ord, name)))
        net = ref_vgsl.TorchVGSLModel(vgsl=spec)
        net.eval()
        # give GroupNorm a non-trivial affine and biases non-zero values
        peephole = {k.rsplit('.weight_ip_l0', 1)[0] for k in net.state_dict() if '.weight_ip_l0' in k}
        for k, v in net.state_dict().items():
            if 'Gn' in k or k.endswith('bias'):
                v.copy_(torch.randn(v.shape) * 0.5 + (1.0 if k.endswith('layer.weight') else 0.0))
            elif any(k.startswith(pfx + '.') for pfx in peephole):
                v.copy_(torch.randn(v.shape) * 0.12)    # PeepholeBidiLSTM leaves its parameters uninitialised (layers.py:155-162)
        _, c, h, _ = net.input
        x = torch.randn(n, c, h or 24, w)      # variable-height specs ([1,0,0,1 ...]) get 24 rows
        for k, v in net.state_dict().items():
            out[f'{name}/sd/{k}'] = v.numpy()
        out[f'{name}/vgsl'] = np.array(net.user_metadata['vgsl'])     # the named spec the reference stores with a model
        out[f'{name}/x'] = x.numpy()
        if lens is None:
            y, _ = net.nn(x, None)
            out[f'{name}/y'] = y.numpy()
            if probe and probe[0]:      # the seq_lens the reference returns for this batch (the output itself is not kept)
                _, ol = net.nn(x, torch.tensor(probe[0]))
                out[f'{name}/olens_probe'] = ol.numpy().astype(np.int32)
        else:
            # parity target for ragged batches = each line on its own (batch 1, lens None)
            olens = []
            for i, L in enumerate(lens):
                y, _ = net.nn(x[i:i + 1, ..., :L].contiguous(), None)
                out[f'{name}/y{i}'] = y.numpy()
                olens.append(y.shape[3])
            try:
                _, ol = net.nn(torch.nn.functional.pad(x, (0, 0)), torch.tensor(lens))
            except Exception:       # the peephole cell has no packed form: the reference cannot run it with seq_lens at all
                ol = None
            if ol is not None:
                out[f'{name}/olens'] = ol.numpy().astype(np.int32)
    np.savez_compressed(path, **out)
    print('wrote', path, len(out), 'arrays')


def codec_fixture(path):
    c2l = {'a': [1], 'b': [2], 'c': [3, 4], 'de': [5], 'xyz': [6, 7, 8], 'ܐ': [9]}
    codec = RefCodec(c2l)
    seqs = [
        [(1, 0, 1, 0.5), (2, 2, 3, 0.25), (3, 4, 5, 0.75), (4, 6, 9, 0.25), (5, 10, 12, 1.0)],
        [(6, 0, 1, 0.3), (7, 2, 3, 0.6), (8, 4, 5, 0.9), (9, 6, 7, 0.1)],
        [(3, 0, 1, 0.3), (1, 2, 3, 0.6)],          # dangling first half of a multi-label code
        [(99, 0, 1, 0.3), (1, 2, 3, 0.6)],         # unknown label is skipped
        [],
    ]
    out = {'c2l': json.dumps(c2l), 'seqs': json.dumps(seqs),
           'decoded': json.dumps([[(c, int(s), int(e), float(u)) for c, s, e, u in codec.decode(s_)] for s_ in seqs]),
           'encode_in': json.dumps(['abc', 'dexyzq', 'cdeܐ']),
           'encode_out': json.dumps([codec.encode(s).tolist() for s in ['abc', 'dexyzq', 'cdeܐ']])}
    np.savez_compressed(path, **out)
    print('wrote', path)


def _ref_transforms(batch, height, width, channels, pad, valid_norm):
    """ImageInputTransforms (kraken/lib/dataset/utils.py:93-152) evaluated with the reference's own
    lineest / resize code; torchvision is absent here, so the v2 ops it would call are spelled out with
    PIL + numpy exactly as torchvision defines them (Grayscale = PIL convert('L'), Pad(fill=255),
    PILToTensor, ToDtype(scale=True) = /255)."""
    from PIL import Image, ImageOps
    from kraken.lib import functional_im_transforms as F_t
    from kraken.lib.lineest import CenterNormalizer

    def run(im: Image.Image):
        mode = 'RGB' if channels == 3 else 'L'
        scale = (height, width)
        center = valid_norm and channels == 1 and height > 1 and width == 0
        im = im.convert(mode)
        if scale != (0, 0):
            if center:
                im = F_t.pil_dewarp(im, lnorm=CenterNormalizer(scale[0]))
                im = im.convert(mode)
            else:
                im = F_t.pil_fixed_resize(im, scale=scale)
        if pad:
            im = ImageOps.expand(im, border=(pad, 0), fill=255 if mode == 'L' else (255, 255, 255))   # v2.Pad(fill=255): every channel
        t = torch.from_numpy(np.array(im, dtype=np.uint8))
        t = t[None] if t.dim() == 2 else t.permute(2, 0, 1)
        t = t.to(torch.float32) / 255.0
        return t.max() - t
    return run


@torch.inference_mode()
def overfit_fixture(path):
    from PIL import Image
    from kraken_amd.io import read_coreml
    import kraken.rpred as ref_rpred
    from kraken.containers import BBoxLine, Segmentation

    meta, sd = read_coreml(os.path.join(RES, 'overfit.mlmodel'))
    kwargs = dict(meta)
    net = ref_vgsl.TorchVGSLModel(**kwargs)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    net.eval()
    rec = RefRecognizer(net, device='cpu')
    rec.kind = 'vgsl'

    # ---- reproduce the reference's known-answer test through its own rpred
    class _TS:
        def __init__(self, batch, height, width, channels, pad, valid_norm=True, **kw):
            self.fn = _ref_transforms(batch, height, width, channels, pad[0] if isinstance(pad, tuple) else pad, valid_norm)

        def __call__(self, im):
            return self.fn(im)
    ref_rpred.ImageInputTransforms = _TS
    im = Image.open(os.path.join(RES, '000236.png'))
    bbox = [0, 0, 2544, 156]
    seg = Segmentation(type='bbox', text_direction='horizontal-lr', imagename='000236.png',
                       lines=[BBoxLine(id='foo', bbox=bbox)], script_detection=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        # tests/test_rpred.py:352-358: rpred(model, im, seg, True) -- the 4th positional is `pad` (True == 1)
        pred = list(ref_rpred.rpred(rec, im, seg, True))
        # tests/test_rpred.py:453-462: mm_rpred(..., bidi_reordering=False), default pad 16
        pred_nobidi = list(ref_rpred.mm_rpred(defaultdict(lambda: rec), im, seg, bidi_reordering=False))
    print('rpred bbox pad=1 bidi  :', repr(pred[0].prediction))
    print('rpred bbox pad=16 nobidi:', repr(pred_nobidi[0].prediction))
    assert pred[0].prediction == 'ܡ ܘܡ ܗ ܡܕܐ ܐ ܐܐ ܡ ܗܗܐܐܐܕ', 'known answer tests/test_rpred.py:358 not reproduced'
    assert pred_nobidi[0].prediction == 'ܕܗܣܐܕ ܪܝ .ܡܡ ܐܠܠ ܗܠ ܐܘܗ ܟܘܗܢ ܡܡ ܐܠ', 'known answer tests/test_rpred.py:462 not reproduced'

    box = im.crop(bbox)
    out = {'spec': meta['vgsl'], 'meta': json.dumps({k: v for k, v in meta.items() if k not in ('accuracy', 'metrics')}),
           'string_rpred_pad1_bidi': pred[0].prediction, 'string_rpred_pad16_nobidi': pred_nobidi[0].prediction,
           'cuts_rpred_pad16_nobidi': json.dumps(pred_nobidi[0].cuts),
           'conf_rpred_pad16_nobidi': np.array(pred_nobidi[0].confidences),
           'cuts_rpred_pad1_bidi': json.dumps(pred[0].cuts), 'conf_rpred_pad1_bidi': np.array(pred[0].confidences),
           'bbox': np.array(bbox), 'box_size': np.array(box.size), 'page': np.array(im.convert('L'), dtype=np.uint8)}
    for pad in (1, 16):
        ts = _TS(1, 30, 0, 1, (pad, 0), True)(box)
        logits, _ = net.nn(ts.unsqueeze(0))
        probs = logits.softmax(1).squeeze(2)
        dec = ref_greedy(probs)
        flat, counts = tuples_to_arr(dec)
        chars = net.codec.decode(dec[0])
        out.update({f'pad{pad}_line': ts.numpy(), f'pad{pad}_logits': logits.squeeze(2).numpy(),
                    f'pad{pad}_probs': probs.numpy(), f'pad{pad}_tuples': flat, f'pad{pad}_counts': counts,
                    f'pad{pad}_decoded': json.dumps([(c, int(s), int(e), float(u)) for c, s, e, u in chars]),
                    f'pad{pad}_string_display': ''.join(c for c, *_ in chars)})
    for k, v in sd.items():
        out[f'sd/{k}'] = v.numpy()
    np.savez_compressed(path, **out)
    print('wrote', path, 'line tensor', tuple(ts.shape), 'T', logits.shape[-1])


@torch.inference_mode()
def overfit_models_fixture(path):
    """
    The three other recognisers among the reference's test resources (SURVEY.md 8c list item 1): overfit_newpoly.mlmodel,
    overfit_bl.safetensors, overfit_bl_newpoly.safetensors.  Their own tests run them on baseline segmentations, whose
    polygon extraction needs packages absent here; the network path is pinned instead: the fixture line through BOTH
    transform branches (dewarp = bbox lines, fixed-height resize = what a baseline crop gets) -> the reference's logits,
    softmax, greedy tuples and strings.
    """
    from PIL import Image
    from kraken_amd.io import read_coreml, read_safetensors
    im = Image.open(os.path.join(RES, '000236.png'))
    box = im.crop([0, 0, 2544, 156])
    out = {'files': json.dumps(['overfit_newpoly.mlmodel', 'overfit_bl.safetensors', 'overfit_bl_newpoly.safetensors'])}
    for fi, fname in enumerate(json.loads(out['files'])):
        p = os.path.join(RES, fname)
        meta, sd = read_coreml(p) if fname.endswith('.mlmodel') else read_safetensors(p)[0]
        kwargs = {k: v for k, v in meta.items() if k not in ('_model', 'model_type')}
        net = ref_vgsl.TorchVGSLModel(**kwargs)
        own = net.state_dict()
        missing, unexpected = net.load_state_dict({k: v.to(own[k].dtype) for k, v in sd.items()}, strict=False)
        assert not missing and not unexpected, (fname, missing, unexpected)
        net.eval()
        b, c, h, w = net.input
        out[f'm{fi}_spec'] = kwargs['vgsl']
        out[f'm{fi}_meta'] = json.dumps({k: v for k, v in kwargs.items() if k not in ('accuracy', 'metrics', 'hyper_params')}, default=str)
        for k, v in sd.items():
            out[f'm{fi}_sd/{k}'] = v.float().numpy()
        for vn in (True, False):
            ts = _ref_transforms(b, h, w, c, 16, vn)(box)
            logits, _ = net.nn(ts.unsqueeze(0))
            probs = logits.softmax(1).squeeze(2)
            dec = ref_greedy(probs)
            flat, counts = tuples_to_arr(dec)
            chars = net.codec.decode(dec[0])
            tag = f'm{fi}_{"dewarp" if vn else "resize"}'
            out.update({f'{tag}_line': ts.numpy(), f'{tag}_logits': logits.squeeze(2).numpy(), f'{tag}_probs': probs.numpy(),
                        f'{tag}_tuples': flat, f'{tag}_counts': counts, f'{tag}_string': ''.join(ch for ch, *_ in chars)})
            print(fname, tag, tuple(ts.shape), 'T', logits.shape[-1], repr(out[f'{tag}_string'][:40]))
    np.savez_compressed(path, **out)
    print('wrote', path)


def transforms_fixture(path):
    from PIL import Image
    rng = np.random.RandomState(7)
    out = {}
    cases = []
    for i, (h, w, height, valid_norm, pad) in enumerate([(60, 400, 48, True, 16), (37, 250, 48, False, 16),
                                                         (48, 300, 48, True, 0), (90, 333, 30, True, 16)]):
        arr = np.full((h, w), 255, np.uint8)
        # dark "strokes" on a light page with a wandering baseline
        yc = (h / 2 + 0.15 * h * np.sin(np.arange(w) / 40.0)).astype(int)
        for x in range(0, w, 3):
            if rng.rand() < 0.6:
                lo = max(yc[x] - rng.randint(2, h // 3), 0)
                hi = min(yc[x] + rng.randint(2, h // 3), h)
                arr[lo:hi, x:x + 2] = rng.randint(0, 90)
        im = Image.fromarray(arr, 'L')
        t = _ref_transforms(1, height, 0, 1, pad, valid_norm)(im)
        out[f'im{i}'] = arr
        out[f'out{i}'] = t.numpy()
        cases.append({'height': height, 'valid_norm': valid_norm, 'pad': pad})
    # round 3: 3-channel models (no dewarp: fixed-height LANCZOS resize of an RGB crop) -- up- and down-scaling, own generator so
    # that the cases above stay bit-identical
    rng3 = np.random.RandomState(11)
    for h, w, height, pad in [(61, 410, 48, 16), (33, 257, 48, 16), (48, 120, 48, 8)]:
        i = len(cases)
        arr = rng3.randint(120, 256, size=(h, w, 3)).astype(np.uint8)
        for x in range(0, w, 5):
            if rng3.rand() < 0.5:
                arr[h // 4:3 * h // 4, x:x + 2] = rng3.randint(0, 100, size=3)
        im = Image.fromarray(arr, 'RGB')
        t = _ref_transforms(1, height, 0, 3, pad, False)(im)
        out[f'im{i}'] = arr
        out[f'out{i}'] = t.numpy()
        cases.append({'height': height, 'valid_norm': False, 'pad': pad, 'channels': 3})
    out['cases'] = json.dumps(cases)
    np.savez_compressed(path, **out)
    print('wrote', path)


def random_group_cases(n=14, seed=11):
    """Randomly nested small networks the reference accepts (the generator of spec_names_fixture, another seed): numeric goldens
    for arbitrary nesting, next to the hand-written GROUP_CASES."""
    import random
    rng = random.Random(seed)

    def leaf():
        return rng.choice(['Cr3,3,%d' % rng.choice([4, 8]), 'Cl1,1,%d' % rng.choice([4, 8]), 'Ct3,5,%d' % rng.choice([4, 8]), 'Do', 'I', 'Gn2'])

    def series(depth):
        return '[' + ' '.join(block(depth - 1) for _ in range(rng.randint(1, 3))) + ']'

    def parallel(depth):
        return '(' + ' '.join((rng.choice([leaf(), series(depth - 1)]) if depth > 0 else leaf()) for _ in range(rng.randint(2, 3))) + ')'

    def block(depth):
        if depth <= 0:
            return leaf()
        r = rng.random()
        return leaf() if r < 0.35 else (series(depth) if r < 0.6 else parallel(depth))

    cases = {}
    while len(cases) < n:
        body = ' '.join(block(3) for _ in range(rng.randint(1, 3)))
        tail = rng.choice([' S1(1x0)1,3 Lbx8 O1c5', ' Mp2,2 S1(1x0)1,3 Lfx6', ' O2l3', ' Cr3,3,4'])
        spec = f'[1,12,0,{rng.choice([1, 2, 3])} Cr3,3,4 {body}{tail}]'
        if '(' not in spec:
            continue
        try:
            ref_vgsl.TorchVGSLModel(vgsl=spec)
        except Exception:
            continue
        w = rng.choice([19, 26, 33])
        cases[f'rnd{len(cases):02d}'] = (spec, 2, w, [w, w - 7] if rng.random() < 0.5 else None)
    return cases


def random_forms_cases(n=32, seed=23):
    """Random sequential networks mixing the round-5 forms -- general reshapes, Additions over every axis but the batch, transposed
    convolutions -- with convolutions, pools, GroupNorm and, where the height ends at 1, a recurrent tail: whatever the reference
    builds AND runs becomes a golden (tensors compared without seq_lens: a reshape's seq_lens rule is pinned elsewhere)."""
    import random
    rng = random.Random(seed)

    def dims(spec):
        m = ref_vgsl.TorchVGSLModel(vgsl=spec + ']')
        return m.output[1], m.output[2]

    def divisors(v):
        return [a for a in range(1, v + 1) if v % a == 0]

    cases, tries = {}, 0
    while len(cases) < n and tries < 4000:
        tries += 1
        spec = f'[1,{rng.choice([8, 12])},0,{rng.choice([1, 2, 3])} C{rng.choice("rtl")}3,3,{rng.choice([4, 6, 8])}'
        used = set()
        try:
            for _ in range(rng.randint(2, 5)):
                c, h = dims(spec)
                kind = rng.choice(['conv', 'conv', 'pool', 'ct', 'sc2h', 'sh2c', 'sh2w', 'add', 'gn'])
                if kind == 'conv':
                    blk = f'C{rng.choice("rtl")}{rng.choice([1, 3])},{rng.choice([3, 5])},{rng.choice([4, 6, 8])}'
                elif kind == 'pool':
                    blk = 'Mp2,2'
                elif kind == 'ct':
                    blk = rng.choice([f'CTr3,3,{rng.choice([4, 8])},2,2', f'CTl2,2,{rng.choice([4, 6])},2,2', f'CTt3,3,{rng.choice([4, 6])}'])
                elif kind == 'sc2h':
                    a = rng.choice(divisors(c))
                    blk = rng.choice([f'S3({a}x{c // a})1,3', f'S3({a}x{c // a})3,1', f'S3({a}x0)1,3'])
                elif kind == 'sh2c':
                    a = rng.choice(divisors(h))
                    blk = rng.choice([f'S1({a}x{h // a})1,3', f'S1({a}x{h // a})3,1', f'S1(0x{h // a})3,1'])
                elif kind == 'sh2w':
                    a = rng.choice(divisors(h))
                    blk = rng.choice([f'S1({a}x{h // a})1,2', f'S1({a}x{h // a})2,1'])
                elif kind == 'add':
                    ax = rng.choice([1, 3, 2])
                    size = {1: h, 3: c, 2: 9}[ax]
                    blk = f'A{ax},{rng.randint(1, size)}'
                else:
                    g = rng.choice(divisors(c))
                    blk = f'Gn{g}'
                used.add(kind)
                spec += ' ' + blk
            c, h = dims(spec)
            if h == 1 and rng.random() < 0.7:
                spec += rng.choice([' Lbx8 O1c5', ' Lfx6', ' O1c4'])
            elif rng.random() < 0.4 and h <= 16:
                spec += ' S1(1x0)1,3 Lbx8 O1c5'
            spec += ']'
            if not used & {'ct', 'sc2h', 'sh2c', 'sh2w', 'add'}:
                continue
            net = ref_vgsl.TorchVGSLModel(vgsl=spec)
            net.eval()
            w = rng.choice([17, 24, 31])
            with torch.no_grad():
                y, _ = net.nn(torch.randn(2, net.input[1], net.input[2], w), None)
            if y.numel() == 0 or y.numel() > 200000:
                continue
        except Exception:
            continue
        cases[f'rf{len(cases):02d}'] = (spec, 2, w, None)
    return cases


def spec_names_fixture(path, n=80, seed=7):
    """Randomly nested specs (serial / parallel groups to depth 3, Addition-free leaves that keep H and W) through the reference's
    parser: the state-dict keys + shapes, the named spec and the output shape it derives -- or the fact that it refuses the spec.
    tests/test_host_cpu.py::test_random_nested_specs_parse_like_the_reference compares kraken_amd's parser against them."""
    import random
    rng = random.Random(seed)

    def leaf():
        return rng.choice(['Cr3,3,%d' % rng.choice([4, 8, 12]), 'Cl1,1,%d' % rng.choice([4, 8]), 'Ct3,5,%d' % rng.choice([4, 8]),
                           'Do', 'Do0.2', 'I', 'Gn2'])

    def series(depth):
        return '[' + ' '.join(block(depth - 1) for _ in range(rng.randint(1, 3))) + ']'

    def parallel(depth):
        return '(' + ' '.join((rng.choice([leaf(), series(depth - 1)]) if depth > 0 else leaf()) for _ in range(rng.randint(2, 3))) + ')'

    def block(depth):
        if depth <= 0:
            return leaf()
        r = rng.random()
        return leaf() if r < 0.4 else (series(depth) if r < 0.7 else parallel(depth))

    out = []
    for _ in range(n):
        body = ' '.join(block(3) for _ in range(rng.randint(1, 4)))
        tail = rng.choice(['', ' S1(1x0)1,3 Lbx8 O1c5', ' Mp2,2 S1(1x0)1,3 Lfx6', ' O2l3'])
        spec = f'[1,12,0,{rng.choice([1, 2, 3])} Cr3,3,4 {body}{tail}]'
        try:
            m = ref_vgsl.TorchVGSLModel(vgsl=spec)
            out.append({'spec': spec, 'ok': True, 'keys': {k: list(v.shape) for k, v in m.state_dict().items()},
                        'vgsl': m.user_metadata['vgsl'], 'output': list(m.output)})
        except Exception as e:
            out.append({'spec': spec, 'ok': False, 'error': type(e).__name__})
    with open(path, 'w') as f:
        json.dump(out, f, indent=0)
    print('wrote', path, len(out), 'specs,', sum(r['ok'] for r in out), 'accepted by the reference')


def reshape_random_fixture(path, n=240, seed=11):
    """Random Reshape / Addition layers through the reference (layers.py:188-223, 285-335; build_reshape / build_addition,
    model.py:616-635, 739-777): every axis as the split axis, every target, fixed parts and `-1` parts, fixed and variable widths, the
    batch axis.  Per case: the spec, the call's (N, W), whether the reference builds it (else the exception type), the output of an
    arange tensor (a reshape only moves values: exact integers) and the seq_lens it returns for a ragged length vector.
    tests compare the parser (construction errors, static shape), both oracles and -- on the GPU -- the permuted-copy kernel."""
    import random
    rng = random.Random(seed)
    meta, arrays = [], {}

    def factor(v):
        cands = [(a, v // a) for a in range(1, v + 1) if v % a == 0]
        a, b = rng.choice(cands)
        r = rng.random()
        if r < 0.25:
            return 0, b             # the first part left to the tensor
        if r < 0.5:
            return a, 0
        if r < 0.58:
            return a + 1, b         # does not divide: torch's reshape raises
        return a, b

    for i in range(n):
        N, C, H, W = rng.choice([1, 2, 4, 6]), rng.choice([2, 3, 4, 6, 8]), rng.choice([1, 2, 3, 4, 6]), rng.choice([4, 6, 8, 12])
        var_w = rng.random() < 0.4
        if rng.random() < 0.2:        # Addition, incl. the batch axis
            dim = rng.choice([0, 1, 2, 3])
            size = {0: N, 1: H, 2: W, 3: C}[dim]
            chunk = rng.choice([c for c in range(1, size + 1)] + [size + 1])
            block = f'A{dim},{chunk}'
        else:
            src = rng.choice([0, 1, 2, 3])
            size = {0: N, 1: H, 2: W, 3: C}[src]
            a, b = factor(size)
            other = rng.choice([0, 1, 2, 3])
            high, low = (src, other) if rng.random() < 0.5 else (other, src)
            if rng.random() < 0.05:
                high = (src + 1) % 4
                low = (src + 2) % 4  # neither is the source: ValueError
            block = f'S{src}({a}x{b}){high},{low}'
        spec = f'[{N},{H},{0 if var_w else W},{C} {block}]'
        rec = {'spec': spec, 'n': N, 'w': W, 'i': i}
        try:
            net = ref_vgsl.TorchVGSLModel(vgsl=spec)
        except Exception as e:
            rec.update(ok=False, error=type(e).__name__)
            meta.append(rec)
            continue
        net.eval()
        x = torch.arange(N * C * H * W, dtype=torch.float32).reshape(N, C, H, W)
        rec.update(ok=True, static=list(net.nn[-1].output_shape))
        try:
            with torch.no_grad():
                y, _ = net.nn(x, None)
            rec['runs'] = True
            arrays[f'y{i}'] = y.numpy().astype(np.int32)
            lens = [W] + [rng.randint(1, W) for _ in range(N - 1)]
            with torch.no_grad():
                _, ol = net.nn(x, torch.tensor(lens))
            rec.update(lens=lens, olens=[int(v) for v in ol.tolist()])
        except Exception as e:
            rec.update(runs=False, run_error=type(e).__name__)
        meta.append(rec)
    arrays['cases'] = np.array(json.dumps(meta))
    np.savez_compressed(path, **arrays)
    print('wrote', path, len(meta), 'cases,', sum(r['ok'] for r in meta), 'built,', sum(bool(r.get('runs')) for r in meta), 'ran')


if __name__ == '__main__':
    which = sys.argv[1:] or ['overfit', 'overfit_models', 'bench_a', 'bench_b', 'layers', 'image_lstm', 'x3_networks', 'breadth', 'groups', 'groups_random', 'spec_names', 'codec', 'transforms', 'bench_lines', 'bench_lines_r6', 'big_lstm', 'forms_r5', 'reshape_random', 'forms_random', 'sizes_r6']
    if 'overfit' in which:
        overfit_fixture(os.path.join(HERE, 'overfit.npz'))
    if 'overfit_models' in which:
        overfit_models_fixture(os.path.join(HERE, 'overfit_models.npz'))
    if 'bench_a' in which:
        bench_fixture(BENCH_A, os.path.join(HERE, 'bench_a.npz'))
    if 'bench_b' in which:
        bench_fixture(BENCH_B, os.path.join(HERE, 'bench_b.npz'),
                      cases=(('n4w400', 4, 400, [0, 3]), ('n16w800', 16, 800, [7])))
    if 'bench_lines' in which:
        bench_lines_fixture(os.path.join(HERE, 'bench_lines.npz'))
    if 'big_lstm' in which:
        big_lstm_fixture(os.path.join(HERE, 'big_lstm.npz'))
    if 'bench_lines_r6' in which:
        bench_lines_r6_fixture(os.path.join(HERE, 'bench_lines_r6.npz'), only=os.environ.get('R6_ONLY', '').split(',') if os.environ.get('R6_ONLY') else None)
    if 'layers' in which:
        layer_fixture(os.path.join(HERE, 'layers.npz'))
    if 'image_lstm' in which:
        layer_fixture(os.path.join(HERE, 'image_lstm.npz'), IMAGE_LSTM_CASES)
    if 'x3_networks' in which:
        layer_fixture(os.path.join(HERE, 'x3_networks.npz'), X3_NETWORK_CASES)
    if 'breadth' in which:
        layer_fixture(os.path.join(HERE, 'breadth.npz'), BREADTH_CASES)
    if 'sizes_r6' in which:
        layer_fixture(os.path.join(HERE, 'sizes_r6.npz'), SIZES_R6_CASES)
    if 'groups' in which:
        layer_fixture(os.path.join(HERE, 'groups.npz'), GROUP_CASES)
    if 'forms_r5' in which:
        layer_fixture(os.path.join(HERE, 'forms_r5.npz'), FORMS_R5_CASES)
    if 'groups_random' in which:
        layer_fixture(os.path.join(HERE, 'groups_random.npz'), random_group_cases())
    if 'spec_names' in which:
        spec_names_fixture(os.path.join(HERE, 'spec_names.json'))
    if 'reshape_random' in which:
        reshape_random_fixture(os.path.join(HERE, 'reshape_random.npz'))
    if 'forms_random' in which:
        layer_fixture(os.path.join(HERE, 'forms_random.npz'), random_forms_cases())
    if 'codec' in which:
        codec_fixture(os.path.join(HERE, 'codec.npz'))
    if 'transforms' in which:
        transforms_fixture(os.path.join(HERE, 'transforms.npz'))
