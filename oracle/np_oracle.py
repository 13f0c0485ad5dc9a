"""
ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy (fp32) restatement of the arithmetic of kraken's line-recognition hot path,
written from the reference's semantics, one function per reference call site.  It exists only
to check the HIP kernels: it may be imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` and by nothing else; the product package ``kraken_amd``
never imports it and has no CPU path.

Where the arithmetic lives: the reference delegates it to PyTorch (un-vendored dependency,
``pyproject.toml:43`` ``torch>=2.4.0,<=2.12``; this image: torch 2.10.0) through the wrappers
in ``kraken/lib/vgsl/layers.py``.  The published semantics of torch.nn.Conv2d / MaxPool2d /
GroupNorm / LSTM / Linear / softmax are restated here.

Pinning: ``tests/test_oracle_golden.py`` checks every function below against the golden vectors
in ``tests/golden/*.npz``, which were produced by running the UNMODIFIED reference modules
(``/root/reference/kraken``) on CPU -- generator committed as ``tests/golden/make_golden.py`` --
including the reference's own known-answer strings of ``tests/test_rpred.py:352-462``.
Parity status: pinned (fp32, tolerance 2e-4 on logits; identical label tuples).

Semantics for ragged batches: masked padding (see include/kraken_amd.h) -- equal to the
reference's per-line (batch = 1, lens = None) results, which is what the golden vectors hold.
"""
import math
from itertools import groupby

import numpy as np

F32 = np.float32


# ------------------------------------------------------------------ length propagation
def conv_out_len(L, k, s, d, p):
    """ActConv2D.forward seq_len update, kraken/lib/vgsl/layers.py:858-859 (clamp min 1)."""
    return max(int(math.floor((L + 2 * p - d * (k - 1) - 1) / s + 1)), 1)


def pool_out_len(L, k, s):
    """MaxPool.forward seq_len update, kraken/lib/vgsl/layers.py:387."""
    return int(math.floor((L - (k - 1) - 1) / s + 1))


def _extent(size, k, s, d=1, p=0):
    return int(math.floor((size + 2 * p - d * (k - 1) - 1) / s + 1))


# ------------------------------------------------------------------------------ layers
def _activate(out, act):
    """The non-linearity of ActConv2D (layers.py:808-825); 's' (sigmoid) is skipped in forward (:850-852)."""
    if act == 'r':
        out = np.maximum(out, 0)
    elif act == 't':
        out = np.tanh(out)
    elif act == 'lr':
        out = np.where(out > 0, out, F32(0.01) * out)
    elif act == 'm':                                   # torch.nn.Softmax(dim=1), layers.py:814-816
        e = np.exp(out - out.max(axis=1, keepdims=True)).astype(F32)
        out = e / e.sum(axis=1, keepdims=True, dtype=F32)
    elif act not in ('l', 's'):
        raise NotImplementedError(act)
    return out.astype(F32)


def conv_transpose2d(x, w, b, stride=(1, 1), dilation=(1, 1), act='l'):
    """
    ActConv2D(transposed=True).forward, kraken/lib/vgsl/layers.py:826-834, 842-846: torch.nn.ConvTranspose2d with padding
    ((dh*(kh-1))//2, (dw*(kw-1))//2), no output_size.  Restated as its definition -- every input pixel scatters w * x into the output
    at (y*sh - ph + dy*dh, x*sw - pw + dx*dw) -- not as the zero-insertion + convolution the device runs.
    x (N,Cin,H,W), w (Cin,Cout,kh,kw) -> (N,Cout,(H-1)sh - 2ph + dh(kh-1) + 1, ...).
    """
    x = np.asarray(x, F32)
    w = np.asarray(w, F32)
    N, Cin, H, W = x.shape
    _, Cout, kh, kw = w.shape
    sh, sw = stride
    dh, dw = dilation
    ph, pw = (dh * (kh - 1)) // 2, (dw * (kw - 1)) // 2
    Hf, Wf = (H - 1) * sh + dh * (kh - 1) + 1, (W - 1) * sw + dw * (kw - 1) + 1        # before the padding is cut off
    full = np.zeros((N, Cout, Hf, Wf), F32)
    for dy in range(kh):
        for dx in range(kw):
            full[:, :, dy * dh: dy * dh + (H - 1) * sh + 1: sh, dx * dw: dx * dw + (W - 1) * sw + 1: sw] += \
                np.einsum('co,nchw->nohw', w[:, :, dy, dx], x, optimize=True).astype(F32)
    out = full[:, :, ph:Hf - ph, pw:Wf - pw] + np.asarray(b, F32)[None, :, None, None]
    return _activate(out.astype(F32), act)


def conv2d(x, w, b, stride=(1, 1), dilation=(1, 1), act='l'):
    """
    ActConv2D.forward, kraken/lib/vgsl/layers.py:842-860: torch.nn.Conv2d (cross-correlation) with
    padding ((dh*(kh-1))//2, (dw*(kw-1))//2) (:803) + bias + activation; 's' (sigmoid) is skipped
    in forward (:850-852).  x (N,Cin,H,W), w (Cout,Cin,kh,kw).
    """
    x = np.asarray(x, F32)
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    sh, sw = stride
    dh, dw = dilation
    ph, pw = (dh * (kh - 1)) // 2, (dw * (kw - 1)) // 2
    Ho, Wo = _extent(H, kh, sh, dh, ph), _extent(W, kw, sw, dw, pw)
    xp = np.zeros((N, Cin, H + 2 * ph, W + 2 * pw), F32)
    xp[:, :, ph:ph + H, pw:pw + W] = x
    out = np.zeros((N, Cout, Ho, Wo), F32)
    wmat = w.reshape(Cout, Cin, kh * kw).astype(F32)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, :, dy * dh: dy * dh + (Ho - 1) * sh + 1: sh, dx * dw: dx * dw + (Wo - 1) * sw + 1: sw]
            out += np.einsum('oc,nchw->nohw', wmat[:, :, dy * kw + dx], patch, optimize=True).astype(F32)
    out += np.asarray(b, F32)[None, :, None, None]
    return _activate(out, act)


def maxpool2d(x, kernel, stride):
    """MaxPool.forward, kraken/lib/vgsl/layers.py:381-388: torch.nn.MaxPool2d(k, s), no padding, floor."""
    N, C, H, W = x.shape
    kh, kw = kernel
    sh, sw = stride
    Ho, Wo = _extent(H, kh, sh), _extent(W, kw, sw)
    out = np.full((N, C, Ho, Wo), -np.inf, F32)
    for i in range(kh):
        for j in range(kw):
            out = np.maximum(out, x[:, :, i: i + (Ho - 1) * sh + 1: sh, j: j + (Wo - 1) * sw + 1: sw])
    return out


def groupnorm(x, gamma, beta, groups, lens=None, eps=1e-5):
    """
    GroupNorm.forward, kraken/lib/vgsl/layers.py:967-984: fp32, statistics per (sample, group);
    with seq_len given and any len < W each sample is normalised over its first len columns only
    and the rest stays zero (:979-984).
    """
    x = np.asarray(x, F32)
    N, C, H, W = x.shape
    out = np.zeros_like(x)
    for n in range(N):
        L = W if lens is None else min(max(int(lens[n]), 1), W)
        xs = x[n, :, :, :L].reshape(groups, -1).astype(np.float64)
        mean = xs.mean(axis=1, keepdims=True)
        var = xs.var(axis=1, keepdims=True)
        y = ((xs - mean) / np.sqrt(var + eps)).reshape(C, H, L)
        out[n, :, :, :L] = (y * gamma[:, None, None] + beta[:, None, None]).astype(F32)
    return out


def reshape_hc(x):
    """Reshape.forward for S1(1x0)1,3, kraken/lib/vgsl/layers.py:313-335: (N,C,H,W)->(N,H*C,1,W), feature h*C+c."""
    N, C, H, W = x.shape
    return np.ascontiguousarray(x.transpose(0, 2, 1, 3).reshape(N, H * C, 1, W))


def reshape_general(x, p, lens=None):
    """
    Reshape.forward in general, kraken/lib/vgsl/layers.py:313-333 (axes in NCHW numbering, as build_reshape maps them,
    model.py:758-775): axis `src` is split into part_a x part_b, the list of axes is rotated so that one part lands in front of axis
    `high` / `low`, the two merge.  seq_lens are scaled by (width in front) / (width behind) -- a float32 product, truncated.
    """
    x = np.asarray(x, F32)
    w0, src = x.shape[3], p['src']
    x5 = x.reshape(x.shape[:src] + (p['a'], p['b']) + x.shape[src + 1:])      # (numpy takes -1 like torch)
    dest = p['low']
    if p['high'] != src:
        dest = p['high']
    else:
        src += 1
    perm = list(range(5))
    perm.insert(dest, perm.pop(src))                      # layers.py:323-327 bubbles axis `src` to `dest` by adjacent swaps: this rotation
    x5 = x5.transpose(perm)
    o = np.ascontiguousarray(x5.reshape(x5.shape[:dest] + (x5.shape[dest] * x5.shape[dest + 1],) + x5.shape[dest + 2:]))
    if lens is not None:
        lens = [int(F32(L) * F32(float(w0) / o.shape[3])) for L in lens]
    return o, lens


def _sigmoid(v):
    return (1.0 / (1.0 + np.exp(-v))).astype(F32)


def lstm(x, weights, hidden, direction='b', lens=None, peephole=False):
    """
    TransposedSummarizingRNN.forward, kraken/lib/vgsl/layers.py:513-547, for L{f,r,b}x on an input of
    height 1: torch.nn.LSTM(batch_first, 1 layer) over the width axis with pack_padded_sequence /
    pad_packed_sequence when lens is given.  Gates in torch order i,f,g,o;
    c' = sig(f) c + sig(i) tanh(g); h' = sig(o) tanh(c'); the reverse direction starts at each
    line's own last valid step; padded outputs are 0.
    x (N,C,1,W) -> (N, D*hidden, 1, W).  weights: list per direction of (w_ih, w_hh, b_ih, b_hh).
    peephole: PeepholeLSTMCell, layers.py:72-103 (legacy ocropy models): i and f add w_ip c / w_fp c, the output gate adds w_op c'
    and is NOT squashed -- hy = (o + w_op c') tanh(c'); weights[d] = (w_ih, w_hh, b_ih, (w_ip, w_fp, w_op)).
    """
    N, Cin, Hh, W = x.shape
    assert Hh == 1
    seq = x[:, :, 0, :].transpose(0, 2, 1).astype(F32)   # (N, W, C)
    # NB: the reference builds nn.LSTM(bidirectional = direction == 'b') and never flips the input, so
    # 'r' ("reverse") runs FORWARD exactly like 'f' (layers.py:496-511) -- restated as is.
    dirs = {'f': [False], 'r': [False], 'b': [False, True], 'rev': [True]}[direction]
    out = np.zeros((N, W, hidden * len(dirs)), F32)
    for d, rev in enumerate(dirs):
        if peephole:
            w_ih, w_hh, b_ih = (np.asarray(a, F32) for a in weights[d][:3])
            w_ip, w_fp, w_op = (np.asarray(a, F32) for a in weights[d][3])
            b_hh = np.zeros(4 * hidden, F32)
        else:
            w_ih, w_hh, b_ih, b_hh = (np.asarray(a, F32) for a in weights[d])
        xp = (seq.reshape(-1, Cin) @ w_ih.T + b_ih).reshape(N, W, 4 * hidden).astype(F32)
        for n in range(N):
            L = W if lens is None else int(lens[n])
            h = np.zeros(hidden, F32)
            c = np.zeros(hidden, F32)
            steps = range(L - 1, -1, -1) if rev else range(L)
            for t in steps:
                g = xp[n, t] + (w_hh @ h + b_hh).astype(F32)
                i, f, gg, o = g[:hidden], g[hidden:2 * hidden], g[2 * hidden:3 * hidden], g[3 * hidden:]
                if peephole:
                    c = _sigmoid(f + w_fp * c) * c + _sigmoid(i + w_ip * c) * np.tanh(gg).astype(F32)
                    h = ((o + w_op * c) * np.tanh(c)).astype(F32)
                else:
                    c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(gg).astype(F32)
                    h = _sigmoid(o) * np.tanh(c).astype(F32)
                out[n, t, d * hidden:(d + 1) * hidden] = h
    return np.ascontiguousarray(out.transpose(0, 2, 1))[:, :, None, :]


def lstm_image(x, weights, hidden, direction='b', axis='x', peephole=False):
    """
    TransposedSummarizingRNN.forward on a 4-D image (kraken/lib/vgsl/layers.py:519-547): with axis 'x' every image
    row (n, h) is one sequence over W (NCHW -> HNWC -> (H*N, W, C)); with axis 'y' (`transpose`, :521-523) every
    column (n, w) is one sequence over H.  No seq_lens (the reference raises for height > 1, :528-530).
    x (N,C,H,W) -> (N, D*hidden, H, W).
    """
    N, C, H, W = x.shape
    if axis == 'y':
        return lstm_image(x.transpose(0, 1, 3, 2), weights, hidden, direction, 'x', peephole).transpose(0, 1, 3, 2)
    rows = x.transpose(0, 2, 1, 3).reshape(N * H, C, 1, W)            # one height-1 "line" per image row
    o = lstm(rows, weights, hidden, direction, None, peephole)         # (N*H, D*hidden, 1, W)
    return np.ascontiguousarray(o[:, :, 0, :].reshape(N, H, -1, W).transpose(0, 2, 1, 3))


def linear(x, w, b):
    """LinSoftmax.forward, kraken/lib/vgsl/layers.py:710-722: Linear over the channel axis; logits, no softmax."""
    N, C, H, W = x.shape
    y = np.einsum('oc,nchw->nohw', np.asarray(w, F32), x.astype(F32), optimize=True) + np.asarray(b, F32)[None, :, None, None]
    return y.astype(F32)


def mask_width(x, lens):
    """Masked-padding rule: zero every column >= the line's valid width."""
    if lens is None:
        return x
    x = x.copy()
    for n, L in enumerate(lens):
        x[n, ..., max(int(L), 0):] = 0
    return x


# --------------------------------------------------------------------------- whole network
def forward(specs, sd, x, lens=None):
    """
    MultiParamSequential.forward, kraken/lib/vgsl/layers.py:44-53, with masked-padding semantics.
    `specs`: list of objects with .kind/.name/.params (kraken_amd.vgsl.parse_vgsl output);
    `sd`: state dict {key: ndarray} with the reference's key names.
    Returns (output (N,C,H,W') float32, olens list or None).
    """
    x = np.asarray(x, F32)
    cur = None if lens is None else [int(v) for v in lens]
    x = mask_width(x, cur)
    forks = []                                            # parallel groups being evaluated: [input, input lens, member outputs]
    # Behind a layer that changes the number of lines (Addition / Reshape on the batch axis) the reference's seq_lens still count the
    # INPUT's lines: nothing is masked any more, and the layers that use seq_lens per line fail in the reference (pack_padded_sequence;
    # the masked GroupNorm unless every line is full width, layers.py:977-984)
    n_in = x.shape[0]
    detached = False
    for sp in specs:
        k, p, nm = sp.kind, sp.params, getattr(sp, 'key', sp.name)
        if detached and cur is not None:
            if k == 'rnn' and x.shape[2] == 1 and p.get('axis', 'x') == 'x':
                raise ValueError('seq_lens of another batch size reach a packed LSTM (the reference raises)')
            if k == 'groupnorm' and min(cur) < x.shape[3]:
                raise ValueError('seq_lens of another batch size reach a masked GroupNorm (the reference raises)')
        if k == 'dropout':
            continue                                      # identity in eval, layers.py:433-437
        # MultiParamParallel.forward, layers.py:60-71: every member gets the group's input, the outputs are concatenated on the
        # channel axis, the seq_lens are those the LAST member returned
        if k == 'par_begin':
            forks.append([x, cur, []])
            continue
        if k in ('par_next', 'par_end'):
            forks[-1][2].append(x)
            if k == 'par_next':
                x, cur = forks[-1][0], forks[-1][1]
            else:
                x = np.concatenate(forks.pop()[2], axis=1)
            continue
        if k == 'add':                                    # Addition.forward, layers.py:205-210
            ax, ch = p['axis'], p['chunk']
            nk = x.shape[ax] // ch
            pieces = [np.take(x, range(i * ch, (i + 1) * ch), axis=ax) for i in range(nk)]
            acc = pieces[0].astype(F32)
            for piece in pieces[1:]:
                acc = (acc + piece).astype(F32)
            x = acc
            if x.shape[0] != n_in:
                detached = True
            continue
        if k == 'conv' and p.get('transposed'):
            x = conv_transpose2d(x, sd[f'nn.{nm}.co.weight'], sd[f'nn.{nm}.co.bias'], p['stride'], p['dilation'], p['nl'])
            if cur is not None:       # layers.py:855: (seq_len - 1) stride - 2 padding + dilation (kernel - 1) + 1
                cur = [(L - 1) * p['stride'][1] - 2 * p['padding'][1] + p['dilation'][1] * (p['kernel'][1] - 1) + 1 for L in cur]
        elif k == 'conv':
            x = conv2d(x, sd[f'nn.{nm}.co.weight'], sd[f'nn.{nm}.co.bias'], p['stride'], p['dilation'], p['nl'])
            if cur is not None:
                cur = [conv_out_len(L, p['kernel'][1], p['stride'][1], p['dilation'][1], p['padding'][1]) for L in cur]
        elif k == 'maxpool':
            x = maxpool2d(x, p['kernel'], p['stride'])
            if cur is not None:
                cur = [pool_out_len(L, p['kernel'][1], p['stride'][1]) for L in cur]
        elif k == 'groupnorm':
            x = groupnorm(x, sd[f'nn.{nm}.layer.weight'], sd[f'nn.{nm}.layer.bias'], p['groups'], None if detached else cur)
        elif k == 'reshape' and p.get('general'):
            n0 = x.shape[0]
            x, cur = reshape_general(x, p, cur)
            detached = detached or x.shape[0] != n0
        elif k == 'reshape':
            x = reshape_hc(x)
        elif k == 'rnn':
            ws = []
            for sfx in [''] + (['_reverse'] if p['direction'] == 'b' else []):
                if p.get('legacy') == 'ocropy':
                    w_ih = np.asarray(sd[f'nn.{nm}.layer.weight_ih_l0{sfx}'], F32)
                    ws.append((w_ih[:, 1:], sd[f'nn.{nm}.layer.weight_hh_l0{sfx}'], w_ih[:, 0],
                               tuple(sd[f'nn.{nm}.layer.weight_{g_}p_l0{sfx}'] for g_ in 'ifo')))
                elif p.get('legacy') == 'clstm':
                    # a constant 1 in front of every input vector, no biases (layers.py:498-499, 522-524): the first weight
                    # column acts as the bias
                    w_ih = np.asarray(sd[f'nn.{nm}.layer.weight_ih_l0{sfx}'], F32)
                    ws.append((w_ih[:, 1:], sd[f'nn.{nm}.layer.weight_hh_l0{sfx}'], w_ih[:, 0], np.zeros(w_ih.shape[0], F32)))
                else:
                    ws.append(tuple(sd[f'nn.{nm}.layer.{w}_l0{sfx}'] for w in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')))
            if x.shape[2] != 1 or p.get('axis', 'x') == 'y':
                # rows as sequences + seq_lens: the reference raises (layers.py:528-530); columns as sequences: seq_lens pass
                # through untouched (every column runs its full height) and the width mask below zeroes the padding columns
                assert cur is None or p.get('axis', 'x') == 'y', 'seq_lens with an LSTM over image rows (the reference raises)'
                x = lstm_image(x, ws, p['hidden'], p['direction'], p.get('axis', 'x'), p.get('legacy') == 'ocropy')
                if p.get('summarize'):                      # o[:, :, -1, :].unsqueeze(2), layers.py:537-539
                    x = x[:, :, -1:, :] if p.get('axis', 'x') == 'y' else x[:, :, :, -1:]
            else:
                x = lstm(x, ws, p['hidden'], p['direction'], None if detached else cur, p.get('legacy') == 'ocropy')
                if p.get('summarize'):
                    # the reference raises when a seq_len exceeds the one column that is left (layers.py:543-545)
                    assert cur is None or max(cur) <= 1, 'Do not use summarizing layer in x-axis with batching/sequences'
                    x = x[:, :, :, -1:]
        elif k == 'linear':
            x = linear(x, sd[f'nn.{nm}.lin.weight'], sd[f'nn.{nm}.lin.bias'])
        else:
            raise NotImplementedError(k)
        if k not in ('linear',) and not detached:
            x = mask_width(x, cur)
    return x, cur


# ---------------------------------------------------------------------- softmax + decode
def softmax_c(logits, temperature=1.0):
    """`(logits / T).softmax(1)`, kraken/lib/vgsl/rpred.py:226 and kraken/lib/models.py:115. logits (N,C,T)."""
    z = np.asarray(logits, F32) / F32(temperature)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z).astype(F32)
    return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)


def greedy_decode(outputs, seq_lens=None):
    """
    greedy_decoder, kraken/lib/ctc_decoder.py:35-72: per line argmax/max over classes for the valid
    steps, runs of equal labels collapsed, blank (0) dropped; tuples (label, first step, last step
    INCLUSIVE, max confidence of the run).  Ties resolve to the lowest class index.
    """
    outputs = np.asarray(outputs)
    if outputs.ndim == 2:
        outputs = outputs[None]
    if seq_lens is None:
        if outputs.shape[0] != 1:
            raise ValueError('seq_lens need to be set for batch decoding.')
        seq_lens = [outputs.shape[-1]]
    dec = []
    for seq, L in zip(outputs, seq_lens):
        L = int(L)
        labels = seq[:, :L].argmax(axis=0)
        confs = seq[:, :L].max(axis=0)
        line, t = [], 0
        for lab, grp in groupby(labels.tolist()):
            n = len(list(grp))
            if lab != 0:
                line.append((lab, t, t + n - 1, float(confs[t:t + n].max())))
            t += n
        dec.append(line)
    return dec


def codec_decode(l2c, tuples):
    """PytorchCodec.decode, kraken/lib/codec.py:148-195, for `l2c` = {label tuple: string}."""
    out, i = [], 0
    labs = tuple(int(t[0]) for t in tuples)
    while i < len(labs):
        hit = False
        for key, s in l2c.items():
            if labs[i:i + len(key)] == tuple(key):
                grp = tuples[i:i + len(key)]
                conf = grp[0][3] if len(key) == 1 else float(np.mean([g[3] for g in grp]))
                out.extend((ch, grp[0][1], grp[-1][2], conf) for ch in s)
                i += len(key)
                hit = True
                break
        if not hit:
            i += 1
    return out


# ---------------------------------------------------------------------------------------------------------------------
# CenterNormalizer dewarp (reference kraken/lib/lineest.py:26-87, called through functional_im_transforms.pil_dewarp :50-52) with the
# arithmetic of the scipy.ndimage calls spelled out -- scipy is an un-vendored dependency of the reference (pinned scipy >= 1.13 in its
# pyproject; 1.15.3 in the authoring container), so its published algorithms are restated here and pinned BIT FOR BIT against
# scipy itself (tests/test_oracle_golden.py) and against the reference's outputs in tests/golden/transforms.npz:
#   gaussian_filter   = correlate1d per axis with weights exp(-x^2 / 2 sigma^2) / sum, radius int(4 sigma + 0.5); a symmetric kernel
#                       is summed as  w0 x[c] + sum_{j = r .. 1} w_j (x[c - j] + x[c + j])  (far to near), fp64; integer input ->
#                       integer output by C truncation; boundary 'constant' (0) for the image, 'reflect' for the ridge
#   uniform_filter    = running sum per axis:  t += x[l + size - 1] - x[l - 1];  out = t / size;  window [l - size // 2, ...)
#   affine_transform  = (diagonal matrix, order 1, mode 'constant') bilinear at coordinate o / scale, cval outside [0, n - 1],
#                       weights (wy wx) summed over (0,0) (0,1) (1,0) (1,1) in fp64, result stored as float32
# This is what csrc/dewarp.hip implements on the device.
def _gauss_weights(sigma: float):
    r = int(4.0 * float(sigma) + 0.5)
    x = np.arange(-r, r + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), r


def _correlate_sym(a: np.ndarray, w: np.ndarray, r: int, axis: int, mode: str) -> np.ndarray:
    a = np.moveaxis(np.asarray(a, dtype=np.float64), axis, -1)
    n = a.shape[-1]
    if mode == 'constant':
        ext = np.concatenate([np.zeros(a.shape[:-1] + (r,)), a, np.zeros(a.shape[:-1] + (r,))], axis=-1)
    else:                                       # 'reflect': (d c b a | a b c d | d c b a), repeated for r > n
        idx = np.arange(-r, n + r)
        idx = np.mod(idx, 2 * n)
        idx = np.where(idx >= n, 2 * n - 1 - idx, idx)
        ext = a[..., idx]
    out = ext[..., r:r + n] * w[r]
    for j in range(r, 0, -1):                   # far to near, pairs first
        out = out + (ext[..., r - j:r - j + n] + ext[..., r + j:r + j + n]) * w[r - j]
    return np.moveaxis(out, -1, axis)


def _uniform_1d(a: np.ndarray, size: int, axis: int) -> np.ndarray:
    a = np.moveaxis(np.asarray(a, dtype=np.float64), axis, -1)
    n = a.shape[-1]
    s1 = size // 2
    s2 = size - s1 - 1
    ext = np.concatenate([np.zeros(a.shape[:-1] + (s1,)), a, np.zeros(a.shape[:-1] + (s2,))], axis=-1)
    out = np.empty_like(a)
    t = np.zeros(a.shape[:-1])
    for l in range(size):
        t = t + ext[..., l]
    out[..., 0] = t / size
    for l in range(1, n):
        t = t + (ext[..., l + size - 1] - ext[..., l - 1])
        out[..., l] = t / size
    return np.moveaxis(out, -1, axis)


def line_centers_np(ink: np.ndarray, smoothness: float = 1.0, extra: float = 0.3) -> np.ndarray:
    """lineest.CenterNormalizer.measure's centre line (lineest.py:34-44)."""
    h, w = ink.shape
    w0, r0 = _gauss_weights(h * 0.5)
    w1, r1 = _gauss_weights(h * smoothness)
    blur = _correlate_sym(_correlate_sym(ink, w0, r0, 0, 'constant'), w1, r1, 1, 'constant')
    uni = _uniform_1d(_uniform_1d(blur, int(h * 0.5), 0), int(w), 1)
    blur = blur + 0.001 * uni
    ridge = np.argmax(blur, axis=0)
    w2, r2 = _gauss_weights(h * extra)
    smooth = _correlate_sym(ridge, w2, r2, 0, 'reflect')
    return smooth.astype(np.int64).astype(np.int32)            # integer output array: C truncation


def center_normalize_np(gray: np.ndarray, target_height: int, spread: float = 4.0) -> np.ndarray:
    """lineest.dewarp + CenterNormalizer.measure/normalize (lineest.py:26-87) on a grayscale line (0 ink .. 255 paper, float)."""
    line = np.asarray(gray, dtype=np.float64)
    top = np.amax(line)
    ink = top - line
    ink = ink * 1.0 / np.amax(ink)
    h, w = ink.shape
    center = line_centers_np(ink)
    rows = np.arange(h)[:, None]
    mad = np.mean(np.abs(rows - center[None, :])[ink != 0])
    r = int(1 + spread * mad)
    stack = np.vstack([top * np.ones((h, w)), line, top * np.ones((h, w))])
    mid = center + h
    band = np.array([stack[mid[i] - r:mid[i] + r, i] for i in range(w)], dtype=np.float32).T
    bh, bw = band.shape
    scale = target_height * 1.0 / bh
    z = 1.0 / scale
    oh, ow = target_height, int(scale * bw)
    b = band.astype(np.float64)
    cy = np.arange(oh) * z
    cx = np.arange(ow) * z
    out = np.full((oh, ow), float(top))
    iy = np.nonzero((cy >= 0) & (cy <= bh - 1))[0]
    ix = np.nonzero((cx >= 0) & (cx <= bw - 1))[0]
    y0 = np.floor(cy[iy]).astype(int)
    x0 = np.floor(cx[ix]).astype(int)
    ty = (cy[iy] - y0)[:, None]
    tx = (cx[ix] - x0)[None, :]
    pad = np.full((bh + 1, bw + 1), float(top))                # neighbours past the last sample carry weight 0 (value irrelevant)
    pad[:bh, :bw] = b
    acc = pad[y0[:, None], x0[None, :]] * ((1 - ty) * (1 - tx))
    acc = acc + pad[y0[:, None], x0[None, :] + 1] * ((1 - ty) * tx)
    acc = acc + pad[y0[:, None] + 1, x0[None, :]] * (ty * (1 - tx))
    acc = acc + pad[y0[:, None] + 1, x0[None, :] + 1] * (ty * tx)
    out[np.ix_(iy, ix)] = acc
    return out.astype(np.float32)
