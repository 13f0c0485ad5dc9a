"""
VGSL model object whose ``nn(x, seq_lens)`` operator runs on hand-written HIP
kernels for gfx950 (MI355X).

Mirrors the public surface of the reference's ``kraken.lib.vgsl.TorchVGSLModel``
(kraken/lib/vgsl/model.py:78-568) for the recognition hot path:

* constructor ``TorchVGSLModel(vgsl=spec, codec=..., **metadata)`` (model.py:109-200);
* layer names ``C_0, Mp_2, L_12, O_18 ...`` with a global running index (model.py:53-64), which
  are the state-dict keys ``nn.<name>.co.weight`` etc. -- reference
  weight files load unchanged;
* ``.nn(x, seq_lens) -> (logits (N,C,1,T), olens)`` (model.py:488-489, layers.py:44-53);
* ``.input/.output/.codec/.user_metadata/.one_channel_mode/.seg_type/.model_type/
  .use_legacy_polygons/.hyper_params`` (model.py:343-398);
* ``init_weights`` (model.py:450-479), ``add_codec`` (:481-486), ``forward`` (:488-489).

torch modules are used only as parameter containers (device memory, state dict); all
arithmetic of the forward is in ``csrc/*.hip`` behind the C ABI of ``include/kraken_amd.h``.
There is no CPU execution path: calling ``nn`` without a HIP device raises.
"""
import ctypes as C
import json
import math
import re
from dataclasses import dataclass, field
from typing import Any, Optional, Sequence

import logging

import numpy as np
import torch
from torch import nn

from . import _lib
from .codec import PytorchCodec

__all__ = ['TorchVGSLModel', 'parse_vgsl', 'LayerSpec', 'HipSequential']


# --------------------------------------------------------------------------- spec parsing
@dataclass
class LayerSpec:
    kind: str                      # conv | maxpool | groupnorm | dropout | reshape | rnn | linear | add | par_begin | par_next | par_end
    name: str                      # state-dict name, e.g. 'C_0' ('' for the par_* markers)
    text: str                      # spec block with the {name} inserted (named_spec entry; '' for the par_* markers)
    params: dict = field(default_factory=dict)
    in_shape: tuple = ()           # (batch, channels, height, width); 0 = variable
    out_shape: tuple = ()
    # containers around the layer, outermost first: (module name, 'series' | 'parallel').  The reference registers a nested
    # `[ ... ]` / `( ... )` block under the space-joined names of the layers inside it (model.py:236), which makes the
    # state-dict key of a nested layer 'nn.<names of the block>.<...>.<name>.co.weight'
    path: tuple = ()

    @property
    def key(self) -> str:
        """State-dict prefix below 'nn.': container names and the layer's own name, dot-joined."""
        return '.'.join([c for c, _ in self.path] + [self.name])

    @property
    def structural(self) -> bool:
        return self.kind in ('par_begin', 'par_next', 'par_end')


_RE_INPUT = re.compile(r'^(\d+),(\d+),(\d+),(\d+)$')
_RE_NAME = r'(?:\{(?P<name>\w+)\})?'
_GRAMMAR = [
    ('conv', re.compile(r'^C(?P<trans>T)?(?P<nl>lr|s|t|r|l|m)' + _RE_NAME +
                        r'(?P<ky>\d+),(?P<kx>\d+),(?P<out>\d+)(?:,(?P<sy>\d+),(?P<sx>\d+))?(?:,(?P<dy>\d+),(?P<dx>\d+))?$')),
    ('maxpool', re.compile(r'^Mp' + _RE_NAME + r'(?P<ky>\d+),(?P<kx>\d+)(?:,(?P<sy>\d+),(?P<sx>\d+))?$')),
    ('groupnorm', re.compile(r'^Gn' + _RE_NAME + r'(?P<groups>\d+)$')),
    ('dropout', re.compile(r'^Do' + _RE_NAME + r'(?P<p>\d+(?:\.\d*)?|\.\d+)?(?:,(?P<dim>\d+))?$')),
    ('reshape', re.compile(r'^S' + _RE_NAME + r'(?P<dim>\d+)\((?P<a>\d+)x(?P<b>\d+)\)(?P<high>\d+),(?P<low>\d+)$')),
    ('rnn', re.compile(r'^(?P<cell>L|G)(?P<dir>f|r|b)(?P<axis>x|y)(?P<sum>s)?(?P<legacy>c|o)?' + _RE_NAME +
                       r'(?P<out>\d+)$')),
    ('output', re.compile(r'^O' + _RE_NAME + r'(?P<dim>[012])(?P<type>l|s|c)(?P<aug>a)?(?P<out>\d+)$')),
    ('identity', re.compile(r'^I' + _RE_NAME + r'$')),
    ('add', re.compile(r'^A' + _RE_NAME + r'(?P<dim>\d+),(?P<chunk>\d+)$')),
]
_TYPE_TAG = {'conv': 'C', 'maxpool': 'Mp', 'groupnorm': 'Gn', 'dropout': 'Do', 'reshape': 'S', 'output': 'O', 'identity': 'I',
             'add': 'A'}


def reshape_shape(shape: Sequence[int], p: dict) -> tuple:
    """
    Reshape.forward (reference layers.py:313-330) on a shape: axis p['src'] is split into a x b (one may be -1), the reference's
    rotation moves one part in front of axis `high` / `low`, the two merge.  Raises like torch's reshape when the parts do not
    divide the axis.
    """
    src, a, b = p['src'], p['a'], p['b']
    size = shape[src]
    if (a == -1 and (b < 1 or size % b)) or (b == -1 and (a < 1 or size % a)) or (a > 0 and b > 0 and a * b != size):
        want = list(shape[:src]) + [a, b] + list(shape[src + 1:])
        raise RuntimeError(f"shape '{want}' is invalid for input of size {int(np.prod(shape))}")
    if a == -1:
        a = size // b
    elif b == -1:
        b = size // a
    d5 = list(shape[:src]) + [a, b] + list(shape[src + 1:])
    dest, s_ = p['low'], src
    if p['high'] != src:
        dest = p['high']
    else:
        s_ += 1
    # the part at position s_ of the 5-D view travels to position `dest`, the axes in between close the gap (the reference bubbles
    # it there by adjacent swaps: the same rotation), then positions dest and dest + 1 merge
    pd = list(d5)
    pd.insert(dest, pd.pop(s_))
    return tuple(pd[:dest] + [pd[dest] * pd[dest + 1]] + pd[dest + 2:])


def _floor_out(size: int, k: int, s: int, d: int = 1, p: int = 0) -> int:
    """Output extent of a conv / pool window along one axis (0 stays variable)."""
    if size == 0:
        return 0
    return int(math.floor((size + 2 * p - d * (k - 1) - 1) / s + 1))


def _named_block(block: str, name: str) -> str:
    """'Cr3,13,32' + 'C_0' -> 'Cr{C_0}3,13,32' (what the reference stores in named_spec)."""
    block = re.sub(r'\{.+\}', '', block)
    m = re.match(r'^[^\d]+', block)
    head = m.group(0) if m else ''
    return f'{head}{{{name}}}{block[len(head):]}'


def _parse_layer(block: str, shape: tuple, idx: int) -> LayerSpec:
    """One VGSL layer block -> LayerSpec (reference model.py:570-817, one build_* method per kind)."""
    if not block:
        raise ValueError(' invalid layer definition')
    hit = None
    for kind, rx in _GRAMMAR:
        mm = rx.match(block)
        if mm:
            hit = (kind, mm)
            break
    if hit is None:
        if re.match(r'^W', block):
            raise NotImplementedError(f'VGSL block "{block}" is not supported by the HIP executor')
        raise ValueError(f'{block} invalid layer definition')
    kind, mm = hit
    g = mm.groupdict()
    tag = g['cell'] if kind == 'rnn' else _TYPE_TAG[kind]
    if kind == 'output' and g['dim'] == '2':
        tag = g['type']   # heatmap heads are named after their type letter: 'l_8' (reference model.py:811)
    name = g.get('name') or f'{tag}_{idx}'
    n, c, h, w = shape
    p: dict[str, Any] = {}
    if kind == 'conv':
        ky, kx, out = int(g['ky']), int(g['kx']), int(g['out'])
        sy, sx = (int(g['sy']), int(g['sx'])) if g['sx'] else (1, 1)
        dy, dx = (int(g['dy']), int(g['dx'])) if g['dx'] else (1, 1)
        p = dict(kernel=(ky, kx), out=out, stride=(sy, sx), dilation=(dy, dx), nl=g['nl'],
                 padding=((dy * (ky - 1)) // 2, (dx * (kx - 1)) // 2))
        if g['trans']:
            # ActConv2D(transposed=True) (layers.py:826-834): ConvTranspose2d with the same padding rule; its smallest output size
            # (get_shape, layers.py:864-880, no target shape): (in - 1) s - 2 p + d (k - 1) + 1, 0 for a variable axis
            p['transposed'] = True
            up = lambda v, k, s_, d, pd: (v - 1) * s_ - 2 * pd + d * (k - 1) + 1 if v else 0     # noqa: E731
            oshape = (n, out, up(h, ky, sy, dy, p['padding'][0]), up(w, kx, sx, dx, p['padding'][1]))
        else:
            oshape = (n, out, _floor_out(h, ky, sy, dy, p['padding'][0]), _floor_out(w, kx, sx, dx, p['padding'][1]))
    elif kind == 'maxpool':
        ky, kx = int(g['ky']), int(g['kx'])
        sy, sx = (int(g['sy']), int(g['sx'])) if g['sx'] else (ky, kx)
        p = dict(kernel=(ky, kx), stride=(sy, sx))
        oshape = (n, c, _floor_out(h, ky, sy), _floor_out(w, kx, sx))
    elif kind == 'groupnorm':
        p = dict(groups=int(g['groups']))
        oshape = shape
    elif kind == 'dropout':
        p = dict(p=float(g['p']) if g['p'] else 0.5, dim=int(g['dim']) if g['dim'] else 1)
        oshape = shape
    elif kind == 'add':
        # layers.Addition (layers.py:188-223, model.py:616-635): the axis is cut into pieces of `chunk` entries which are summed,
        # out[j] = sum_k in[k * chunk + j] (what is left over behind the last whole piece is dropped: Tensor.unfold)
        dim, chunk = int(g['dim']), int(g['chunk'])
        if dim > 3:
            raise ValueError(f'Invalid dimension {dim} in addition block')
        axis = {0: 0, 1: 2, 2: 3, 3: 1}[dim]
        # (over the batch axis the output has `chunk` lines and the seq_lens, handed through, still count the input's lines)
        # (checked against channels and height only: the spec's batch size is not the call's, and a width in the shape arithmetic
        # may be the 1 the reference's get_shape puts in for a variable dim behind a Reshape -- the call's tensor decides there)
        if chunk < 1 or (axis in (1, 2) and shape[axis] and chunk > shape[axis]):
            raise ValueError(f'addition "{block}": chunk size {chunk} does not fit an axis of {shape[axis]} entries')
        p = dict(axis=axis, chunk=chunk)
        oshape = tuple(chunk if a == axis else v for a, v in enumerate(shape))
    elif kind == 'identity':     # layers.Identity (model.py:637-650): elided like dropout
        kind, p, oshape = 'dropout', dict(identity=True), shape
    elif kind == 'reshape':
        src, a, b, high, low = int(g['dim']), int(g['a']), int(g['b']), int(g['high']), int(g['low'])
        if src != high and src != low:
            raise ValueError(f'Either high ({high}) or low ({low}) must be source dimension ({src})')
        if a == 0 and b == 0:
            raise ValueError('Only one size may be -1')
        if a == 0:
            a = -1
        elif b == 0:
            b = -1
        if src == 1 and high == 1 and low == 3 and h != 0 and ((a == 1 and b in (-1, h)) or (a == -1 and b == h)):
            # the form on the recognition path: fold height into channels, S1(1x0)1,3 (feature index h*C + c): fused into the
            # convolution in front of it / one pass into sequence rows.  Round 6: the same collapse spelled with the height written
            # out -- S1(1x12)1,3, the form of kraken's classic recognition specs (48 rows, two 2x2 pools) -- is the same operation on
            # the same tensor and takes the same path (it ran as a general permuted copy, which kept such recognisers off the
            # pipelined engine and the split-bf16 kernels)
            p = dict(src=src, a=1, b=-1, high=high, low=low)
            # the reference derives this shape from a dummy tensor with variable dims set to 1
            oshape = (n or 1, c * h, 1, w or 1)
        else:
            # every other form (Reshape.forward, layers.py:313-333; build_reshape, model.py:739-777): a permuted copy.  Axes in NCHW
            # numbering from here on; the output shape like get_shape's: the forward arithmetic on a tensor whose variable dims are 1
            dim_map = {0: 0, 1: 2, 2: 3, 3: 1}
            p = dict(general=True, src=dim_map[src], a=a, b=b, high=dim_map[high], low=dim_map[low])
            oshape = reshape_shape(tuple(v or 1 for v in shape), p)
    elif kind == 'rnn':
        # 'G' parses as a GRU but the reference builds the same torch.nn.LSTM for it (layers.py:504-511, model.py:579-593):
        # an alias, layer name G_<idx>
        # 'c': the clstm layout -- a constant 1 in front of every input vector instead of biases (layers.py:498-499, 522-524,
        # nn.LSTM(bias=False)): the first weight column IS the bias; folded when the plan is compiled.  'o': ocropy's peephole cell
        # (layers.py:72-186, selected at :506-507): i and f look at c, the output gate at the new c and is NOT squashed; always
        # bidirectional, no biases, the same constant 1 in front of the input
        hidden = int(g['out'])
        if g['legacy'] == 'o' and g['dir'] != 'b':
            # the reference builds PeepholeBidiLSTM whatever the direction letter and then fails to reshape its 2 x hidden outputs
            raise ValueError(f'RNN variant "{block}": the ocropy peephole cell is bidirectional (the reference fails on any other direction)')
        # axis 'y' = the reference's `transpose`: image columns are the sequences (layers.py:521-523);
        # 's' keeps only the last step of every column (:537-539): (N, C, H, W) -> (N, O, 1, W)
        # on the x axis 's' keeps the last COLUMN: (N, C, H, W) -> (N, O, H, 1) (get_shape, :549-561)
        p = dict(hidden=hidden, direction=g['dir'], cell=g['cell'], axis=g['axis'], summarize=bool(g['sum']),
                 legacy={'c': 'clstm', 'o': 'ocropy'}.get(g['legacy']))
        oc = hidden * (2 if g['dir'] == 'b' else 1)
        oshape = (n, oc, h, w) if not g['sum'] else ((n, oc, 1, w) if g['axis'] == 'y' else (n, oc, h, 1))
    else:  # output
        dim, typ, out = int(g['dim']), g['type'], int(g['out'])
        if dim == 0:
            raise ValueError('categorical output not supported, yet.')
        if typ == 'c' and dim == 2:
            raise ValueError('CTC not supported for heatmap output')
        # (a heatmap head ignores the 'a' flag: build_output reads it on the LinSoftmax branch only, model.py:806-816)
        if dim == 2:
            # 1x1 ActConv2D: 'l' with the (skipped) sigmoid, 's' with a softmax over the classes (reference model.py:806-811)
            kind = 'conv'
            p = dict(kernel=(1, 1), out=out, stride=(1, 1), dilation=(1, 1), nl='s' if typ == 'l' else 'm', padding=(0, 0),
                     output_type=typ)
        else:
            kind = 'linear'
            # 'a': LinSoftmax(augmentation=True) prepends a constant 1 to every input vector (layers.py:703-719); the extra
            # weight column is folded into the bias when the plan is compiled
            p = dict(out=out, output_type=typ, aug=bool(g['aug']))
        oshape = (n, out, h, w)
    return LayerSpec(kind, name, _named_block(block, name), p, shape, oshape)


def _depth_change(block: str, opening: str, closing: str, other_open: str, other_close: str) -> int:
    """
    The reference's `_bracket_count` / `_parenthesis_count` (model.py:819-845): leading `opening` characters (runs of the
    other kind of opening bracket are looked through) minus trailing `closing` characters of one block.
    """
    n = 0
    for c in block:
        if c == opening:
            n += 1
        elif c != other_open:
            break
    for c in reversed(block):
        if c == closing:
            n -= 1
        elif c != other_close:
            break
    return n


class _SpecParser:
    """
    Recursive descent over the space-separated blocks of a spec, following the reference's `_parse` / `build_series` /
    `build_parallel` (model.py:202-241, 847-905): a block starting with '[' opens a serial group, one starting with '('
    a parallel group whose members all see the group's input and whose outputs are concatenated on the channel axis
    (layers.py:56-71); layer indices run globally over the leaf layers, groups take none.  The result is a FLAT list:
    leaf LayerSpecs with their container path, and par_begin / par_next / par_end markers around the members of a
    parallel group -- the order krk_plan_create consumes.
    """

    def __init__(self):
        self.idx = -1

    def parse(self, shape: tuple, blocks: Sequence[str], parallel: bool = False):
        specs: list[LayerSpec] = []
        names: list[str] = []
        i = 0
        first_out = None
        channels = 0
        oshape = shape
        while i < len(blocks):
            block = blocks[i]
            if block and block[0] in '[(':
                series = block[0] == '['
                o, c = ('[', ']') if series else ('(', ')')
                oo, oc = ('(', ')') if series else ('[', ']')
                if block[-1] == c:          # single block in brackets
                    inner, used = [block[1:-1]], 1
                else:
                    depth, used = 0, 0
                    for used, b in enumerate(blocks[i:]):
                        depth += _depth_change(b, o, c, oo, oc)
                        if depth == 0:
                            break
                    if depth:
                        raise ValueError('Unbalanced parentheses in VGSL spec')
                    inner = [block[1:]] + list(blocks[i + 1:i + used]) + [blocks[i + used][:-1]]
                    used += 1
                sub, sub_names, oshape = self.parse(shape, inner, parallel=not series)
                if not sub_names:
                    raise ValueError(f'{block} invalid layer definition')
                cname = (' '.join(sub_names), 'series' if series else 'parallel')
                leaves = [sp for sp in sub if not sp.structural]
                for sp in sub:
                    sp.path = (cname,) + sp.path
                leaves[0].text = o + leaves[0].text
                leaves[-1].text = leaves[-1].text + c
                if not series:
                    sub = ([LayerSpec('par_begin', '', '', {}, shape, shape, (cname,))] + sub +
                           [LayerSpec('par_end', '', '', {}, shape, oshape, (cname,))])
                if len(sub_names) != used:
                    # the reference advances by the number of LAYERS a group returned (model.py:235), which equals the blocks
                    # it spans for every spec it accepts
                    raise ValueError(f'{block} invalid layer definition')
            else:
                self.idx += 1
                try:
                    sp = _parse_layer(block, shape, self.idx)
                except Exception:
                    self.idx -= 1
                    raise
                sub, sub_names, oshape, used = [sp], [sp.name], sp.out_shape, 1
            if parallel:
                if first_out is not None and first_out[2:] != oshape[2:]:
                    raise ValueError('Output shape in parallel block not equal!')
                if first_out is not None:
                    specs.append(LayerSpec('par_next', '', '', {}, shape, shape))
                first_out = first_out or oshape
                channels += oshape[1]
            else:
                shape = oshape
            specs += sub
            names += sub_names
            i += used
        if parallel and first_out is not None:
            oshape = (first_out[0], channels) + tuple(first_out[2:])
        return specs, names, oshape


def parse_vgsl(spec: str):
    """
    Parses a VGSL spec into the input 4-tuple (batch, channels, height, width) and a flat list of LayerSpec (see
    _SpecParser for nested `[ ... ]` / `( ... )` groups).  Grammar: SURVEY.md Appendix A (reference model.py:570-905).
    Raises ValueError for malformed specs and NotImplementedError for valid VGSL the HIP executor does not cover
    (transposed / softmax convolutions, legacy RNN cells, addition over batch or width, wav2vec masking, general reshapes).
    """
    spec = spec.strip()
    if not spec or spec[0] != '[' or spec[-1] != ']':
        raise ValueError('Non-sequential models not supported')
    blocks = spec[1:-1].split(' ')
    m = _RE_INPUT.match(blocks[0])
    if not m:
        raise ValueError('Invalid input spec.')
    batch, height, width, channels = (int(v) for v in m.groups())
    layers, _, _ = _SpecParser().parse((batch, channels, height, width), blocks[1:])
    return (batch, channels, height, width), layers


# ------------------------------------------------------------------ parameter containers
logger = logging.getLogger(__name__)

# Module/attribute names below are part of the on-disk format (state-dict keys).
class _ConvHolder(nn.Module):
    def __init__(self, spec: LayerSpec):
        super().__init__()
        p = spec.params
        conv = nn.ConvTranspose2d if p.get('transposed') else nn.Conv2d
        self.co = conv(spec.in_shape[1], p['out'], p['kernel'], stride=p['stride'], padding=p['padding'], dilation=p['dilation'])


class _GroupNormHolder(nn.Module):
    def __init__(self, spec: LayerSpec):
        super().__init__()
        self.layer = nn.GroupNorm(spec.params['groups'], spec.in_shape[1])


class _PeepholeParams(nn.Module):
    """Parameter holder with the names of the reference's PeepholeBidiLSTM (layers.py:146-170): per direction weight_ih / weight_hh and
    the three peephole vectors weight_ip / weight_fp / weight_op; no biases."""

    def __init__(self, input_size: int, hidden: int):
        super().__init__()
        for sfx in ('', '_reverse'):
            for name, shape in (('weight_ih', (4 * hidden, input_size)), ('weight_hh', (4 * hidden, hidden)),
                                ('weight_ip', (hidden,)), ('weight_fp', (hidden,)), ('weight_op', (hidden,))):
                setattr(self, f'{name}_l0{sfx}', nn.Parameter(torch.zeros(shape)))


class _RnnHolder(nn.Module):
    def __init__(self, spec: LayerSpec):
        super().__init__()
        legacy = spec.params.get('legacy') is not None
        if spec.params.get('legacy') == 'ocropy':
            self.layer = _PeepholeParams(spec.in_shape[1] + 1, spec.params['hidden'])
            return
        self.layer = nn.LSTM(spec.in_shape[1] + (1 if legacy else 0), spec.params['hidden'],
                             bidirectional=spec.params['direction'] == 'b', batch_first=True, bias=not legacy)


class _LinearHolder(nn.Module):
    def __init__(self, spec: LayerSpec):
        super().__init__()
        self.lin = nn.Linear(spec.in_shape[1] + (1 if spec.params.get('aug') else 0), spec.params['out'])


class _NoParams(nn.Module):
    pass


class _Group(nn.Module):
    """A nested `[ ... ]` (MultiParamSequential) or `( ... )` (MultiParamParallel, layers.py:39-71) block: a parameter-free
    container whose children carry the reference's module names, so that the state-dict keys agree."""

    def __init__(self, kind: str):
        super().__init__()
        self.kind = kind

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i: int):
        return list(self._modules.values())[i]


_HOLDERS = {'conv': _ConvHolder, 'groupnorm': _GroupNormHolder, 'rnn': _RnnHolder, 'linear': _LinearHolder}
_ACTS = {'l': _lib.ACT_LINEAR, 'r': _lib.ACT_RELU, 't': _lib.ACT_TANH, 'lr': _lib.ACT_LEAKY, 's': _lib.ACT_SIGMOID,
         'm': _lib.ACT_SOFTMAX}
# The reference builds nn.LSTM(bidirectional = direction == 'b') and never flips the sequence, so an
# 'r' layer runs forward like 'f' (kraken/lib/vgsl/layers.py:496-511); mirrored here on purpose.
_DIRS = {'f': _lib.DIR_FWD, 'r': _lib.DIR_FWD, 'b': _lib.DIR_BIDI}


def _f32(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.detach().to('cpu', torch.float32).numpy())


def layer_table(specs: Sequence[LayerSpec], modules: 'HipSequential'):
    """
    The `krk_layer` table of a network (include/kraken_amd.h) and the host arrays its weight pointers refer to (keep them alive until
    krk_plan_create has returned).  No device needed: INTEGRATION.md section 3's in-tree binding builds the same table from the
    reference's own modules, and a CPU test compares the two.
    """
    descs, keep = [], []
    for spec in specs:
        if spec.kind == 'dropout':
            continue   # identity in eval mode (reference layers.py:433-437)
        d = _lib.KrkLayer()
        if spec.structural:
            d.op = {'par_begin': _lib.OP_PAR_BEGIN, 'par_next': _lib.OP_PAR_NEXT, 'par_end': _lib.OP_PAR_END}[spec.kind]
            descs.append(d)
            continue
        mod = modules.holder(spec)
        arrays: list[np.ndarray] = []
        p = spec.params
        if spec.kind == 'conv':
            d.op = _lib.OP_CONV
            d.cout = p['out']
            d.kh, d.kw = p['kernel']
            d.sh, d.sw = p['stride']
            d.dh, d.dw = p['dilation']
            d.act = _ACTS[p['nl']]
            arrays = [_f32(mod.co.weight), _f32(mod.co.bias)]
            if p.get('transposed'):       # the kernel of the equivalent convolution: (in, out) swapped, both spatial axes flipped
                d.op = _lib.OP_CONVT
                arrays[0] = np.ascontiguousarray(arrays[0].transpose(1, 0, 2, 3)[:, :, ::-1, ::-1])
        elif spec.kind == 'maxpool':
            d.op = _lib.OP_MAXPOOL
            d.kh, d.kw = p['kernel']
            d.sh, d.sw = p['stride']
        elif spec.kind == 'groupnorm':
            d.op = _lib.OP_GROUPNORM
            d.cout = p['groups']
            arrays = [_f32(mod.layer.weight), _f32(mod.layer.bias)]
        elif spec.kind == 'reshape' and p.get('general'):
            d.op = _lib.OP_RESHAPE                         # include/kraken_amd.h: axes in NCHW numbering
            d.kh, d.kw, d.sh, d.sw, d.dh = p['src'], p['a'], p['b'], p['high'], p['low']
            d.cout, d.dw = spec.out_shape[1], spec.out_shape[2]
        elif spec.kind == 'reshape':
            d.op = _lib.OP_RESHAPE_HC
        elif spec.kind == 'add':
            d.op = _lib.OP_ADD
            d.kh = {1: 0, 2: 1, 3: 2, 0: 3}[p['axis']]     # include/kraken_amd.h: 0 = channels, 1 = height, 2 = width, 3 = batch
            d.cout = p['chunk']
        elif spec.kind == 'rnn':
            d.op = _lib.OP_LSTM
            d.cout = p['hidden']
            d.direction = _DIRS[p['direction']]
            d.kw = 1 if p.get('axis', 'x') == 'y' else 0   # include/kraken_amd.h: time axis of an LSTM layer
            d.kh = 1 if p.get('summarize') else 0          # include/kraken_amd.h: keep only the last step (L?ys / L?xs)
            sfx = [''] + (['_reverse'] if p['direction'] == 'b' else [])
            for s in sfx:
                w_ih, w_hh = _f32(getattr(mod.layer, f'weight_ih_l0{s}')), _f32(getattr(mod.layer, f'weight_hh_l0{s}'))
                if p.get('legacy') == 'ocropy':   # the same folded 1, and the peephole vectors (i, f, o) in the slot of b_hh
                    d.act = 1                     # include/kraken_amd.h: an LSTM layer with act = 1 is the ocropy peephole cell
                    arrays += [np.ascontiguousarray(w_ih[:, 1:]), w_hh, np.ascontiguousarray(w_ih[:, 0]),
                               np.concatenate([_f32(getattr(mod.layer, f'weight_{g_}p_l0{s}')) for g_ in 'ifo'])]
                elif p.get('legacy'):    # x' = [1, x], no biases: gates = W[:, 0] + W[:, 1:] x + W_hh h
                    arrays += [np.ascontiguousarray(w_ih[:, 1:]), w_hh, np.ascontiguousarray(w_ih[:, 0]),
                               np.zeros(w_ih.shape[0], np.float32)]
                else:
                    arrays += [w_ih, w_hh, _f32(getattr(mod.layer, f'bias_ih_l0{s}')), _f32(getattr(mod.layer, f'bias_hh_l0{s}'))]
        elif spec.kind == 'linear':
            d.op = _lib.OP_LINEAR
            d.cout = p['out']
            w, b = _f32(mod.lin.weight), _f32(mod.lin.bias)
            if p.get('aug'):       # y = W[:, 0] * 1 + W[:, 1:] x + b
                w, b = np.ascontiguousarray(w[:, 1:]), (b + w[:, 0]).astype(np.float32)
            arrays = [w, b]
            if spec.in_shape[2] != 1:
                # LinSoftmax on an image of more than one row (layers.py:710-722: the features of every pixel through the same
                # Linear): a 1x1 convolution without activation, an image again
                d.op = _lib.OP_CONV
                d.kh = d.kw = d.sh = d.sw = d.dh = d.dw = 1
                d.act = _lib.ACT_LINEAR
                arrays[0] = np.ascontiguousarray(w.reshape(w.shape[0], w.shape[1], 1, 1))
        else:
            raise NotImplementedError(f'layer kind {spec.kind} is not supported by the HIP executor')
        for i, a in enumerate(arrays):
            d.w[i] = a.ctypes.data
        keep.append(arrays)
        descs.append(d)
    return descs, keep


class _Plan:
    """Owns one ``krk_plan`` handle (weights repacked + uploaded by the C library)."""

    def __init__(self, specs: Sequence[LayerSpec], modules: 'HipSequential', in_channels: int, in_height: int,
                 device: int, precision: int = _lib.PREC_F32):
        lib = _lib.load()
        _lib.require_gpu()
        descs, keep = layer_table(specs, modules)
        arr = (_lib.KrkLayer * len(descs))(*descs)
        handle = C.c_void_p()
        _lib.check(lib.krk_plan_create(arr, len(descs), in_channels, in_height, precision, device, C.byref(handle)))
        self.handle = handle
        self.device = device
        self._lib = lib
        # plans with a wide recurrent layer on the split-bf16 kernels run the cluster kernel (lstm_ws), whose only failure signal
        # is the status word (krk_plan_status); the others have nothing to report and nn(x) need not synchronise for it
        self.has_status = bool(lib.krk_plan_has_exchange(handle))

    def clone(self) -> '_Plan':
        """A plan of its own over the SAME packed device weights (krk_plan_clone): own workspace, events and status word, no repack,
        no upload -- what the engine's further slots take."""
        handle = C.c_void_p()
        _lib.check(self._lib.krk_plan_clone(self.handle, C.byref(handle)))
        twin = object.__new__(_Plan)
        twin.handle, twin.device, twin._lib, twin.has_status = handle, self.device, self._lib, self.has_status
        return twin

    def out_shape(self, W: int):
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.krk_plan_out_shape(self.handle, W, C.byref(c), C.byref(h), C.byref(w)))
        return c.value, h.value, w.value

    def out_dims(self, N: int, W: int):
        """(lines, channels, height, width) of the output for a batch of N lines of width W (an Addition / Reshape on the batch axis
        changes the number of lines)."""
        n, c, h, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.krk_plan_out_dims(self.handle, N, W, C.byref(n), C.byref(c), C.byref(h), C.byref(w)))
        return n.value, c.value, h.value, w.value

    def olens(self, lens: np.ndarray, W: int = 0) -> np.ndarray:
        """seq_lens behind the network; W = the batch's width (a general Reshape scales them by the batch's widths, layers.py:331-332)."""
        out = np.empty_like(lens)
        _lib.check(self._lib.krk_plan_olens_w(self.handle, lens.ctypes.data, len(lens), W, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.krk_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# kraken.registry.PRECISIONS -> plan (see HipSequential.precision_for_config)
PRECISION_OF_CONFIG = {'64-true': 'f32', '32-true': 'bf16x3', '32': 'bf16x3', 'bf16-true': 'bf16x3', 'bf16-mixed': 'bf16x3',
                       '16-true': 'bf16x3', '16-mixed': 'bf16x3', 'transformer-engine': 'bf16x3',
                       'transformer-engine-float16': 'bf16x3'}


class DecodedBatch:
    """Host copy of the compact greedy-decode result of one batch (see krk_decode_out)."""

    def __init__(self, labels, starts, ends, confs, counts):
        self.labels, self.starts, self.ends, self.confs, self.counts = labels, starts, ends, confs, counts

    def tuples(self) -> list[list[tuple[int, int, int, float]]]:
        """The list-of-lists of (label, start, end, conf) that greedy_decoder returns."""
        out = []
        lab, st, en, cf = self.labels.tolist(), self.starts.tolist(), self.ends.tolist(), self.confs.tolist()
        for n, k in enumerate(self.counts.tolist()):
            out.append(list(zip(lab[n][:k], st[n][:k], en[n][:k], cf[n][:k])))
        return out


class HipSequential(nn.Module):
    """
    The ``nn`` operator: an ordered container of parameter holders (children named like the
    reference's layers) whose ``forward(x, seq_lens)`` executes the HIP plan.
    Counterpart of MultiParamSequential (reference layers.py:39-53).
    """

    def __init__(self, specs: Sequence[LayerSpec], input_shape):
        super().__init__()
        self._specs = list(specs)
        self._input = tuple(input_shape)
        for spec in specs:
            if spec.structural:
                continue
            parent = self
            for cname, ckind in spec.path:
                if cname not in parent._modules:
                    parent.add_module(cname, _Group(ckind))
                parent = parent._modules[cname]
            parent.add_module(spec.name, _HOLDERS[spec.kind](spec) if spec.kind in _HOLDERS else _NoParams())
        self._plans: dict = {}          # (device, precision, weights version, height) -> _Plan, a few heights at most
        self._sum_x = any(s.kind == 'rnn' and s.params.get('summarize') and s.params.get('axis') == 'x' for s in specs)
        self.precision = _lib.PREC_F32
        self.precision_chosen = False      # set_precision() was called (by the caller or by prepare_for_inference): nobody picks another default

    # -- plan management -------------------------------------------------------------
    def set_precision(self, precision) -> None:
        """
        Arithmetic of the GEMM-shaped layers: 'f32' (exact f32 matrix cores, default), 'bf16x3' (split-bf16 operands on
        the bf16 matrix cores, fp32-class results; conv/LSTM/linear networks only) or, opt-in only, 'bf16' (the same
        kernels with the cross terms dropped: plain bf16 operands, logits ~1e-2 from fp32 -- outside the 1e-3 parity
        gate, never selected by `config.precision`; gate: identical strings on the fixtures).
        """
        table = {'f32': _lib.PREC_F32, 'fp32': _lib.PREC_F32,
                 _lib.PREC_F32: _lib.PREC_F32, 'bf16x3': _lib.PREC_BF16X3, _lib.PREC_BF16X3: _lib.PREC_BF16X3,
                 'bf16': _lib.PREC_BF16, _lib.PREC_BF16: _lib.PREC_BF16}
        if precision in PRECISION_OF_CONFIG:
            precision = self.precision_for_config(precision)
        if precision not in table:
            raise ValueError(f'unknown precision {precision!r}; choose "f32", "bf16x3" or one of kraken\'s precision '
                             f'strings {sorted(PRECISION_OF_CONFIG)}')
        self.precision_chosen = True
        if table[precision] != self.precision:
            self.precision = table[precision]
            self.invalidate()

    def precision_for_config(self, precision: str) -> str:
        """
        kraken's ``config.precision`` (kraken/configs/base.py:65, kraken/registry.py:22) -> arithmetic plan.  Every plan
        keeps fp32-class results (the reduced-precision strings are requests for SPEED, which the split-bf16 plan
        already provides at fp32 accuracy), so the mapping only decides between the two fp32-class plans: 'bf16x3'
        (|d logit| ~1e-5; layers up to and including a network's last GroupNorm run on the exact-f32 cores inside that
        plan, see PlanBuilder::build in csrc/capi.hip) and, for '64-true', the all-f32 plan.
        """
        return PRECISION_OF_CONFIG[precision]

    def _weights_version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _drop_plans(self):
        for plan in self._plans.values():
            plan.close()
        self._plans.clear()

    @property
    def _plan(self) -> Optional[_Plan]:
        """The plan used last (diagnostics / tests)."""
        return next(reversed(self._plans.values()), None)

    def invalidate(self):
        """
        Weights, device or arithmetic changed: drops this module's own plans AND the pipelined engines built on them
        (rpred.py).  A mere change of input height or device index of a direct call only adds a plan (``plan()``): it must
        not take down the engines a running ``LinePipeline`` holds.  An engine a pipeline still holds is closed by that
        pipeline when it lets go of it (``RecognitionEngine.in_use``), never underneath it.
        """
        self._drop_plans()
        for engines in self.__dict__.pop('_engines', {}).values():     # pipelined engines hold their own plans (rpred.py)
            for eng in engines:
                if eng.in_use:
                    eng.stale = True
                else:
                    eng.close()

    def has_layer(self, kind: str) -> bool:
        return any(s.kind == kind for s in self._specs)

    def plan(self, device_index: int, height: Optional[int] = None) -> _Plan:
        # legacy [1,1,0,48]-style inputs keep the line height in the channel axis: C=48, H=1
        _, c, h, _ = self._input
        if height is not None and height != h:
            self._specs_for_height(height)   # raises if the weights do not fit that height
            h = height
        key = (device_index, self.precision, self._weights_version(), h)
        plan = self._plans.get(key)
        if plan is None:
            # in-place weight update or another arithmetic: everything is stale.  Another DEVICE or input height only adds a
            # plan (plans and engines of several devices live side by side)
            if any(k[1:3] != key[1:3] for k in self._plans):
                self.invalidate()
            while len(self._plans) >= 4:                         # variable-height models / devices: a few stay planned
                self._plans.pop(next(iter(self._plans))).close()
            precision = self.precision
            plan = self.new_plan(device_index, h)
            if self.precision != precision:                      # fell back to the exact-f32 plan: the key changes with it
                key = (device_index, self.precision, self._weights_version(), h)
            self._plans[key] = plan
        else:
            self._plans[key] = self._plans.pop(key)              # most recently used last
        return plan

    def new_plan(self, device_index: int, height: Optional[int] = None) -> _Plan:
        """
        A plan of its own for the caller (the engine's slots, `plan()`'s cache), with ONE rule for networks the split-bf16
        kernels do not cover: such a network (or, for variable-height models, such a height) keeps the exact-f32 plan instead
        of failing -- said once, never silently -- whether it is reached through `nn(x)`, `rpred`, `mm_rpred` or a
        `ShardedRecognizer`.
        """
        _, c, h, _ = self._input
        if height is not None:
            h = height
        try:
            return _Plan(self._specs, self, c, h, device_index, self.precision)
        except _lib.KrakenAmdError as e:
            if self.precision == _lib.PREC_F32 or e.code != _lib.KRK_E_UNSUPPORTED:
                raise
            logger.warning(f'the split-bf16 plan does not cover this network ({e}); using the exact-f32 plan')
            self.precision = _lib.PREC_F32
            self.invalidate()
            return _Plan(self._specs, self, c, h, device_index, self.precision)

    def _specs_for_height(self, height: int):
        """
        Shape inference for an input whose height differs from the spec's (a padded page in the segmenter,
        spred.py:253-259: torch modules do not care).  Allowed when every weight-bearing layer keeps its input
        channel count; a height-collapsing reshape does not, and the reference would fail in the next layer too.
        """
        n, c, _, w = self._input
        text = f'[{n},{height},{w},{c} ' + ' '.join(s.text for s in self._specs if s.text) + ']'
        _, specs = parse_vgsl(text)
        for a, b in zip(self._specs, specs):
            if a.kind in ('conv', 'groupnorm', 'rnn', 'linear') and a.in_shape[1] != b.in_shape[1]:
                raise ValueError(f'input height {height} does not fit layer {a.name} of a network built for height '
                                 f'{self._input[2]} ({b.in_shape[1]} input features instead of {a.in_shape[1]})')
        return specs

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return r

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i: int):
        return list(self._modules.values())[i]

    def holder(self, spec: LayerSpec) -> nn.Module:
        """The parameter container of a layer (inside its nested groups)."""
        mod = self
        for cname, _ in spec.path:
            mod = mod._modules[cname]
        return mod._modules[spec.name]

    # -- execution ---------------------------------------------------------------------
    @staticmethod
    def _device_index(x: Optional[torch.Tensor] = None) -> int:
        _lib.require_gpu()
        if x is not None and x.is_cuda:
            return x.device.index if x.device.index is not None else torch.cuda.current_device()
        return torch.cuda.current_device()

    def _prep(self, x: torch.Tensor, seq_lens):
        if x.dim() != 4:
            raise ValueError(f'expected a (N, C, H, W) tensor, got shape {tuple(x.shape)}')
        if x.shape[1] != self._input[1]:
            raise ValueError(f'expected {self._input[1]} input channels, got {x.shape[1]}')
        dev = self._device_index(x)
        xd = x.detach().to(device=f'cuda:{dev}', dtype=torch.float32).contiguous()
        lens = None
        if seq_lens is not None:
            lens = np.ascontiguousarray(torch.as_tensor(seq_lens).detach().cpu().numpy().astype(np.int32))
            if lens.shape != (xd.shape[0],):
                raise ValueError('seq_lens must have one entry per line')
        return dev, xd, lens

    @torch.no_grad()
    def forward(self, x: torch.Tensor, seq_lens: Optional[torch.Tensor] = None, output_shape=None):
        """
        nn(x, seq_lens) -> (output, olens).  For a recogniser the output is the (N, C, 1, T)
        logits tensor (a permuted view of the time-major buffer the kernels write) on the GPU.
        """
        dev, xd, lens = self._prep(x, seq_lens)
        self._check_lens(lens)
        plan = self.plan(dev, xd.shape[2])
        N, _, _, W = xd.shape
        n_out, c, h, w = plan.out_dims(N, W)
        seq_out = self._specs_out_is_seq()
        with torch.cuda.device(dev):
            out = torch.empty((n_out, w, c) if seq_out else (n_out, c, h, w), dtype=torch.float32, device=xd.device)
            stream = torch.cuda.current_stream().cuda_stream

            def run():
                _lib.check(plan._lib.krk_forward(plan.handle, xd.data_ptr(), lens.ctypes.data if lens is not None else None,
                                                 N, W, stream, out.data_ptr()))
            if plan.has_status:
                # the recurrent cluster kernel reports a timed-out exchange through the plan's status word only: callers of
                # nn(x) (custom decoders, the segmenter) must not receive such logits silently.  One retry on the streaming
                # kernel (no exchange) before giving up (_lib.checked_run: the same policy as recognize() and the engine)
                _lib.checked_run(plan.handle, run, torch.cuda.current_stream().synchronize, logger)
            else:
                run()
        olens = None
        if lens is not None:
            olens = torch.from_numpy(plan.olens(lens, W))
        if seq_out:
            out = out.permute(0, 2, 1).unsqueeze(2)
        return out, olens

    def _specs_out_is_seq(self) -> bool:
        """True when the network ends in the time-major sequence layout (after the height collapse)."""
        seq = False
        forks = []
        for spec in self._specs:
            if spec.kind in ('dropout', 'add', 'par_end'):
                if spec.kind == 'par_end':
                    forks.pop()                    # the members agree (the plan refuses a mix)
                continue
            if spec.kind == 'par_begin':
                forks.append(seq)
            elif spec.kind == 'par_next':
                seq = forks[-1]
            elif spec.kind == 'reshape' and spec.params.get('general'):
                seq = False                        # a permuted copy: an NCHW image again
            elif spec.kind == 'linear' and spec.in_shape[2] != 1:
                seq = False                        # ... and so is a linear layer over an image of several rows (a 1x1 convolution)
            elif spec.kind in ('reshape', 'linear'):
                seq = True
            elif spec.kind == 'rnn':
                # an LSTM over the rows/columns of an image (height > 1 or y axis) returns an image again
                seq = seq or (spec.in_shape[2] == 1 and spec.params.get('axis', 'x') == 'x')
            else:
                seq = False
        return seq

    def _check_lens(self, lens) -> None:
        """The reference's run-time refusals that depend on seq_lens (TransposedSummarizingRNN.forward, layers.py:540-545)."""
        if lens is not None and self._sum_x and int(lens.max()) > 1:
            raise Exception('Do not use summarizing layer in x-axis with batching/sequences')

    @torch.no_grad()
    def recognize(self, x: torch.Tensor, seq_lens=None, temperature: float = 1.0, want_logits: bool = False,
                  want_probs: bool = False):
        """
        Fused forward + softmax + CTC best-path decode (krk_recognize).
        Returns (DecodedBatch, olens ndarray, logits or None, probs or None); logits/probs are
        (N, C, T) permuted views on the GPU.
        """
        dev, xd, lens = self._prep(x, seq_lens)
        self._check_lens(lens)
        plan = self.plan(dev, xd.shape[2])
        N, _, _, W = xd.shape
        _, c, h, T = plan.out_dims(N, W)      # (a network that changes the number of lines is refused by krk_recognize below)
        with torch.cuda.device(dev):
            d = xd.device
            i32 = dict(dtype=torch.int32, device=d)
            labels, starts, ends = (torch.empty((N, T), **i32) for _ in range(3))
            confs = torch.empty((N, T), dtype=torch.float32, device=d)
            counts = torch.empty((N,), **i32)
            logits = torch.empty((N, T, c), dtype=torch.float32, device=d) if want_logits else None
            probs = torch.empty((N, T, c), dtype=torch.float32, device=d) if want_probs else None
            olens = np.empty((N,), dtype=np.int32)
            dec = _lib.KrkDecodeOut(labels.data_ptr(), starts.data_ptr(), ends.data_ptr(), confs.data_ptr(),
                                    counts.data_ptr(), T)
            stream = torch.cuda.current_stream().cuda_stream
            host = {}

            def run():
                _lib.check(plan._lib.krk_recognize(plan.handle, xd.data_ptr(),
                                                   lens.ctypes.data if lens is not None else None, N, W,
                                                   float(temperature), stream,
                                                   logits.data_ptr() if want_logits else None,
                                                   probs.data_ptr() if want_probs else None,
                                                   olens.ctypes.data, C.byref(dec)))

            def wait():            # the copies to the host wait for the batch
                host['packed'] = torch.stack([labels, starts, ends, confs.view(torch.int32)]).cpu().numpy()
                host['cnt'] = counts.cpu().numpy()
            # an exchange timeout of the cluster kernel is retried once on the streaming kernel, as in forward() and the engine
            _lib.checked_run(plan.handle, run, wait, logger)
            packed, cnt = host['packed'], host['cnt']
        batch = DecodedBatch(packed[0], packed[1], packed[2], packed[3].view(np.float32), cnt)
        return (batch, olens,
                logits.permute(0, 2, 1) if want_logits else None,
                probs.permute(0, 2, 1) if want_probs else None)


# ------------------------------------------------------------------------------ the model
class TorchVGSLModel(nn.Module):
    """
    Drop-in counterpart of kraken.lib.vgsl.TorchVGSLModel for recognition inference.
    See the module docstring for the mirrored surface.
    """
    _kraken_min_version = '5.0.0'

    def __init__(self, **kwargs) -> None:
        super().__init__()
        if (vgsl := kwargs.pop('vgsl', None)) is None:
            raise ValueError('vgsl specification argument is missing in args.')
        self.spec = vgsl
        self.user_metadata: dict[str, Any] = {'accuracy': [], 'metrics': [], 'seg_type': None,
                                              'one_channel_mode': None, 'model_type': []}
        codec = kwargs.get('codec', None)
        self.user_metadata.update(**kwargs)
        (batch, channels, height, width), specs = parse_vgsl(vgsl)
        self.input = (batch, channels, height, width)
        self.layer_specs = specs
        self.nn = HipSequential(specs, self.input)
        self.output = specs[-1].out_shape if specs else self.input
        self.named_spec = [vgsl.strip()[1:-1].split(' ')[0]] + [s.text for s in specs if s.text]
        self.user_metadata['vgsl'] = '[' + ' '.join(self.named_spec) + ']'
        self.criterion = None
        if specs and specs[-1].params.get('output_type') == 'c':
            self.criterion = nn.CTCLoss(reduction='sum', zero_infinity=True)
        if codec is not None:
            self.add_codec(codec if isinstance(codec, PytorchCodec) else PytorchCodec(codec))
        self.init_weights()
        self.eval()

    # -- kraken.models plugin contract (reference kraken/models/base.py:27-120, model.py:491-546) ---
    def prepare_for_inference(self, config=None):
        """
        Configures the model for inference: ``config`` is kraken's RecognitionInferenceConfig /
        SegmentationInferenceConfig (or any object with the fields read below: ``batch_size``, ``temperature``,
        ``padding``, ``bidi_reordering``, ``device``, ``input_padding``).  Called before every ``predict``
        by kraken's task layer (tasks/recognition.py:87); the HIP plan is (re)built lazily on first use.
        """
        kind = type(config).__name__
        if ('Recognition' in kind and 'recognition' not in self.model_type) or \
           ('Segmentation' in kind and 'segmentation' not in self.model_type):
            raise ValueError(f'{self} is a {self.model_type} model. Got incompatible {kind}.')
        self.eval()
        self._inf_config = config
        dev = getattr(config, 'device', None) or 'cuda'
        if isinstance(dev, (list, tuple)):
            dev = dev[0] if dev else 'cuda'
        if isinstance(dev, int):
            dev = f'cuda:{dev}'
        if str(dev).startswith('cpu') or str(dev) == 'auto':
            dev = 'cuda'                      # this implementation has no CPU path
        self.to(dev)
        precision = getattr(config, 'precision', None)
        if precision is None and not self.nn.precision_chosen:
            precision = '32-true'             # kraken's own default (configs/base.py:65) for a config that does not carry one
        if precision is not None:
            # kraken's strings and 'f32' / 'bf16x3' (reference: Fabric(precision=...), model.py:518-523).  A network the
            # split-bf16 kernels do not cover (odd channel counts ...) keeps the exact-f32 plan instead of failing later.
            self.nn.set_precision(precision)
            if self.nn.precision != _lib.PREC_F32 and self.input[2] > 0:
                p = next(self.parameters())
                try:
                    self.nn.plan(p.device.index if p.device.index is not None else torch.cuda.current_device())
                except _lib.KrakenAmdError:
                    self.nn.set_precision('f32')
        return self

    def predict(self, *args, **kwargs):
        """
        Recognition models: ``predict(im, segmentation)`` -> generator of ocr_records (batched underneath,
        kraken_amd.rpred); segmentation models: ``predict(im)`` -> the heatmap dictionary of
        ``_compute_segmentation_map`` (vectorisation into a Segmentation stays in kraken).
        """
        cfg = getattr(self, '_inf_config', None)
        if 'segmentation' in self.model_type and 'recognition' not in self.model_type:
            from .blla import compute_segmentation_map
            return compute_segmentation_map(self, args[0] if args else kwargs['im'],
                                            input_padding=getattr(cfg, 'input_padding', 0))
        from .rpred import recognition_pred
        im = args[0] if args else kwargs['im']
        segmentation = args[1] if len(args) > 1 else kwargs['segmentation']
        return recognition_pred(self, im, segmentation, cfg)

    # -- initialisation (reference model.py:450-479) ------------------------------------
    def init_weights(self) -> None:
        """LSTM orthogonal (+ forget-gate bias 1), conv U(-0.1, 0.1), linear Xavier-uniform with zero
        bias -- applied module by module in registration order like the reference's ``apply``."""
        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight.data)
                nn.init.constant_(m.bias.data, 0)
            elif isinstance(m, nn.LSTM):
                for p in m.parameters():
                    if p.data.dim() == 2:
                        nn.init.orthogonal_(p.data)
                    else:
                        nn.init.constant_(p.data[len(p) // 4:len(p) // 2], 1.0)
            elif isinstance(m, nn.Conv2d):
                for p in m.parameters():
                    nn.init.uniform_(p.data, -0.1, 0.1)
        self.nn.apply(_init)
        self.nn.invalidate()

    def add_codec(self, codec: PytorchCodec) -> None:
        self.codec = codec
        self.user_metadata['codec'] = json.dumps(self.codec.c2l)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.nn.invalidate()
        return r

    def forward(self, x: torch.Tensor, seq_lens: Optional[torch.Tensor] = None):
        return self.nn(x, seq_lens)

    # -- metadata properties (reference model.py:343-398) -------------------------------
    @property
    def one_channel_mode(self):
        return self.user_metadata['one_channel_mode']

    @one_channel_mode.setter
    def one_channel_mode(self, val):
        if val not in ['1', 'L', None]:
            raise ValueError('one_channel_mode {} is not one of [1, L, None]'.format(val))
        self.user_metadata['one_channel_mode'] = val

    @property
    def model_type(self):
        mt = self.user_metadata.get('model_type', [])
        return [mt] if isinstance(mt, str) else mt

    @model_type.setter
    def model_type(self, val):
        if isinstance(val, str):
            val = [val]
        for v in val:
            if v not in ['recognition', 'segmentation']:
                raise ValueError('model_type {} is not one of [recognition, segmentation]'.format(v))
        self.user_metadata['model_type'] = val

    @property
    def seg_type(self):
        return self.user_metadata.get('seg_type', None)

    @seg_type.setter
    def seg_type(self, val):
        if val not in ['bbox', 'baselines', None]:
            raise ValueError('segmentation type {} is not one of [bbox, baselines, None]'.format(val))
        self.user_metadata['seg_type'] = val

    @property
    def hyper_params(self):
        return self.user_metadata['hyper_params']

    @hyper_params.setter
    def hyper_params(self, val: dict):
        self.user_metadata.setdefault('hyper_params', {}).update(val)

    @property
    def use_legacy_polygons(self):
        return self.user_metadata.get('legacy_polygons', True)

    @use_legacy_polygons.setter
    def use_legacy_polygons(self, val: bool):
        self.user_metadata['legacy_polygons'] = val

    # -- model files ----------------------------------------------------------------------
    @classmethod
    def load_model(cls, path):
        """Loads a CoreML (.mlmodel) or safetensors model file written by kraken."""
        from .io import load_model_file
        return load_model_file(path, cls)
