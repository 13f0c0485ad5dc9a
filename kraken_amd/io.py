"""
Model file readers for kraken's two weight formats, without coremltools.

* safetensors with a ``kraken_meta`` JSON metadata entry -- reference
  kraken/models/writers.py:44-89 (layout) and kraken/models/loaders.py:46-150 (reader):
  every tensor key is ``<uuid>.<state-dict key>``, ``kraken_meta`` maps uuid -> constructor
  kwargs (``_model``, ``_tasks``, ``vgsl``, ``codec``, ...).
* CoreML ``.mlmodel`` protobufs -- reference kraken/lib/vgsl/model.py:270-341 and
  kraken/models/_coreml.py:10-85.  Parsed with a small protobuf wire-format reader; the field
  numbers below are those of Apple's public Model.proto / NeuralNetwork.proto.  CoreML stores
  LSTM gates as separate named matrices; they are re-stacked in torch order i, f, g, o and the
  single CoreML bias goes to ``bias_hh`` with ``bias_ih = 0`` (like _coreml.py:28-49).
"""
import json
import struct
from os import PathLike, fspath

import numpy as np
import torch

__all__ = ['load_model_file', 'read_coreml', 'read_safetensors']


# ------------------------------------------------------------------ protobuf wire format
def _varint(buf, i):
    shift = val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _fields(buf):
    """Yields (field number, wire type, value) of one message; value is int or memoryview."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, i = _varint(buf, i)
        elif wt == 1:
            val, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            val, i = buf[i:i + ln], i + ln
        elif wt == 5:
            val, i = buf[i:i + 4], i + 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield fno, wt, val


def _first(buf, fno, default=None):
    for f, _, v in _fields(buf):
        if f == fno:
            return v
    return default


def _all(buf, fno):
    return [v for f, _, v in _fields(buf) if f == fno]


def _floats(weight_params) -> np.ndarray:
    """WeightParams{1: packed float32, 2: float16 bytes}"""
    if weight_params is None:
        return np.zeros(0, dtype=np.float32)
    chunks = []
    for f, wt, v in _fields(weight_params):
        if f == 1:
            if wt == 2:
                chunks.append(np.frombuffer(bytes(v), dtype='<f4'))
            else:  # unpacked repeated float
                chunks.append(np.array(struct.unpack('<f', bytes(v)), dtype=np.float32))
        elif f == 2 and wt == 2 and len(v):
            chunks.append(np.frombuffer(bytes(v), dtype='<f2').astype(np.float32))
    return np.concatenate(chunks).astype(np.float32) if chunks else np.zeros(0, dtype=np.float32)


def _uints(buf, fno):
    """repeated uint64 (packed or not)"""
    out = []
    for f, wt, v in _fields(buf):
        if f != fno:
            continue
        if wt == 0:
            out.append(v)
        else:
            i = 0
            while i < len(v):
                x, i = _varint(v, i)
                out.append(x)
    return out


def _strip(name: str, suffix: str) -> str:
    return name[:-len(suffix)] if name.endswith(suffix) else name


def _lstm_weights(wp, in_size: int, hidden: int):
    """LSTMWeightParams -> (weight_ih, weight_hh, bias) stacked i, f, g, o."""
    ih = np.concatenate([_floats(_first(wp, k)) for k in (1, 2, 3, 4)]).reshape(4 * hidden, in_size)
    hh = np.concatenate([_floats(_first(wp, k)) for k in (20, 21, 22, 23)]).reshape(4 * hidden, hidden)
    bias = [_floats(_first(wp, k)) for k in (40, 41, 42, 43)]
    bias = np.concatenate([b if b.size else np.zeros(hidden, np.float32) for b in bias])
    return ih, hh, bias


def read_coreml(path):
    """Returns (metadata dict incl. 'vgsl' and 'codec', state dict of torch tensors)."""
    with open(fspath(path), 'rb') as fp:
        buf = memoryview(fp.read())
    desc = _first(buf, 2)
    nnet = _first(buf, 500)
    if desc is None or nnet is None:
        raise ValueError(f'{path} is not a CoreML neural network model')
    user = {}
    meta_msg = _first(desc, 100)
    if meta_msg is not None:
        for entry in _all(meta_msg, 100):
            kv = {f: bytes(v).decode('utf-8') for f, _, v in _fields(entry)}
            user[kv.get(1, '')] = kv.get(2, '')
    meta = json.loads(user.get('kraken_meta', '{}'))
    vgsl = user.get('vgsl', meta.get('vgsl'))
    if vgsl is None:
        raise ValueError('No VGSL spec in model metadata')
    codec = user.get('codec', meta.get('codec', 'null'))
    meta['vgsl'] = vgsl
    meta['codec'] = json.loads(codec) if isinstance(codec, str) else codec
    # older files carry model_type as a plain string
    if isinstance(meta.get('model_type'), str):
        meta['model_type'] = [meta['model_type']]

    sd = {}
    for layer in _all(nnet, 1):
        name = bytes(_first(layer, 1, b'')).decode('utf-8')
        for fno, _, body in _fields(layer):
            if fno == 100:      # ConvolutionLayerParams
                base = _strip(name, '_conv')
                out_c, k_c = _first(body, 1, 0), _first(body, 2, 0)
                ksz = _uints(body, 20) or [3, 3]
                if _first(body, 60, 0):
                    raise NotImplementedError('transposed convolutions are not supported by the HIP executor')
                sd[f'nn.{base}.co.weight'] = torch.from_numpy(
                    _floats(_first(body, 90)).reshape(out_c, k_c, ksz[0], ksz[1]).copy())
                sd[f'nn.{base}.co.bias'] = torch.from_numpy(_floats(_first(body, 91)).copy())
            elif fno == 140:    # InnerProductLayerParams
                base = _strip(name, '_lin')
                in_c, out_c = _first(body, 1, 0), _first(body, 2, 0)
                sd[f'nn.{base}.lin.weight'] = torch.from_numpy(_floats(_first(body, 20)).reshape(out_c, in_c).copy())
                sd[f'nn.{base}.lin.bias'] = torch.from_numpy(_floats(_first(body, 21)).copy())
            elif fno in (420, 430):   # Uni/BiDirectionalLSTMLayerParams
                base = _strip(name, '_transposed')
                in_size, hidden = _first(body, 1, 0), _first(body, 2, 0)
                for d, wp in enumerate(_all(body, 20)):
                    sfx = '_reverse' if d == 1 else ''
                    ih, hh, bias = _lstm_weights(wp, in_size, hidden)
                    sd[f'nn.{base}.layer.weight_ih_l0{sfx}'] = torch.from_numpy(ih.copy())
                    sd[f'nn.{base}.layer.weight_hh_l0{sfx}'] = torch.from_numpy(hh.copy())
                    sd[f'nn.{base}.layer.bias_hh_l0{sfx}'] = torch.from_numpy(bias.copy())
                    sd[f'nn.{base}.layer.bias_ih_l0{sfx}'] = torch.zeros(4 * hidden)
            elif fno == 500:    # CustomLayerParams
                cls_name = bytes(_first(body, 10, b'')).decode('utf-8')
                if cls_name == 'groupnorm':
                    ws = _all(body, 20)
                    sd[f'nn.{name}.layer.weight'] = torch.from_numpy(_floats(ws[0]).copy())
                    sd[f'nn.{name}.layer.bias'] = torch.from_numpy(_floats(ws[1]).copy())
    return meta, sd


def read_safetensors(path):
    """Returns a list of (metadata dict, state dict) -- one per model stored in the file."""
    from safetensors import safe_open
    out = []
    with safe_open(fspath(path), framework='pt') as f:
        md = f.metadata()
        if md is None:
            raise ValueError(f'No model metadata found in {path}.')
        try:
            model_map = json.loads(md.get('kraken_meta', 'null'))
        except json.JSONDecodeError as e:
            raise ValueError(f'Invalid `kraken_meta` JSON in {path}: {e}') from e
        if not isinstance(model_map, dict):
            raise ValueError(f'Invalid `kraken_meta` metadata in {path}: expected object.')
        keys = list(f.keys())
        for prefix, data in model_map.items():
            if not isinstance(data, dict):
                raise ValueError(f'Invalid metadata for model `{prefix}` in {path}.')
            meta = dict(data)
            tasks = meta.pop('_tasks', None) or []
            meta.pop('_kraken_min_version', None)
            meta['_model'] = meta.get('_model')
            meta['model_type'] = tasks
            sd = {k[len(prefix) + 1:]: f.get_tensor(k) for k in keys if k.startswith(prefix + '.')}
            out.append((meta, sd))
    return out


def load_model_file(path, cls=None, tasks=('recognition',)):
    """
    Loads a kraken model file into ``cls`` (default: kraken_amd.vgsl.TorchVGSLModel).
    For safetensors files holding several models the first one matching `tasks` is returned.
    """
    if cls is None:
        from .vgsl import TorchVGSLModel as cls
    p = fspath(path) if isinstance(path, PathLike) else str(path)
    with open(p, 'rb') as fp:
        head = fp.read(16)
    entries = None
    if p.endswith('.safetensors') or (len(head) >= 9 and head[8:9] == b'{'):
        entries = read_safetensors(p)
    else:
        entries = [read_coreml(p)]
    for meta, sd in entries:
        mt = meta.get('model_type') or []
        if tasks and mt and not set(tasks).intersection(mt):
            continue
        kwargs = dict(meta)
        kwargs.pop('_model', None)
        model = cls(**kwargs)
        own = model.state_dict()
        cast = {k: v.to(own[k].dtype) if k in own else v for k, v in sd.items()}
        missing, unexpected = model.load_state_dict(cast, strict=False)
        if missing or unexpected:
            raise RuntimeError(f'Error(s) in loading state_dict from {p}:\n'
                               f'    Missing key(s): {missing}\n    Unexpected key(s): {unexpected}')
        return model
    raise ValueError(f'No model for tasks {tasks} found in {p}')


def load_models(path, tasks=None):
    """
    ``kraken.loaders`` entry point (reference contract kraken/models/loaders.py:27-43):
    ``fn(path, tasks=None) -> list[model]``; raises ValueError when the file is not ours to load so
    that kraken falls through to its next loader.
    """
    try:
        return [load_model_file(path, tasks=tuple(tasks) if tasks else ('recognition',))]
    except (NotImplementedError, RuntimeError, KeyError, IndexError, struct.error) as e:
        raise ValueError(f'{path} is not loadable by the MI355X executor: {e}') from e
