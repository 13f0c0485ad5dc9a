"""
Pins the CPU oracles (oracle/np_oracle.py, oracle/torch_port.py) against golden vectors produced by
the UNMODIFIED reference (tests/golden/make_golden.py).  CPU only.
"""
import json

import numpy as np
import pytest
import torch

from kraken_amd.vgsl import parse_vgsl
from oracle import np_oracle
from oracle.torch_port import CpuRecognizer, greedy_decode as torch_greedy
from tests.helpers import arr_to_tuples, layer_cases, load_golden, synth_input
from tests.specs import BENCH_A, BENCH_B

LAYER_TOL = 2e-5
CASES = layer_cases()
CASES.update(layer_cases('image_lstm.npz'))   # LSTMs over image rows/columns, scaled-down BLLA segmenter
CASES.update(layer_cases('breadth.npz'))      # round 4: 'G' cells, hidden sizes above 256, ...
CASES.update(layer_cases('sizes_r6.npz'))     # round 6: odd hidden sizes, feature counts off 8 / 16, channel counts off 16
CASES.update(layer_cases('groups.npz'))       # round 4: nested [ ] / ( ) groups, Addition, x-axis summarising LSTMs
CASES.update(layer_cases('groups_random.npz'))   # ... and 14 randomly nested networks (identity members, groups inside groups inside groups)
CASES.update(layer_cases('forms_r5.npz'))        # round 5: the forms that were still refused (ocropy peephole cell, ...)
CASES.update(layer_cases('forms_random.npz'))    # ... and 32 random networks mixing them with convolutions, pools, GroupNorm, recurrent tails


def _zero_pad_x(x, lens):
    x = x.copy()
    for i, L in enumerate(lens):
        x[i, ..., L:] = 0
    return x


@pytest.mark.parametrize('name', sorted(CASES))
def test_np_oracle_layers(name):
    c = CASES[name]
    _, specs = parse_vgsl(c['spec'])
    if c['lens'] is None:
        y, _ = np_oracle.forward(specs, c['sd'], c['x'])
        assert y.shape == c['y'].shape
        np.testing.assert_allclose(y, c['y'], atol=LAYER_TOL, rtol=1e-4)
        if 'olens_probe' in c:     # the seq_lens the reference's batched call returns (Reshape's width ratio, batch-changing layers)
            _, olens = np_oracle.forward(specs, c['sd'], c['x'], c['lens_probe'])
            assert list(olens) == c['olens_probe'].tolist()
    else:
        y, olens = np_oracle.forward(specs, c['sd'], _zero_pad_x(c['x'], c['lens']), c['lens'])
        for i, want in enumerate(c['ys']):
            w = want.shape[3]
            assert olens[i] == w or want.shape[3] == 0
            np.testing.assert_allclose(y[i:i + 1, ..., :w], want, atol=LAYER_TOL, rtol=1e-4)
        if c['olens'] is not None:
            assert list(olens) == c['olens'].tolist()


X3_NETS = layer_cases('x3_networks.npz')     # whole small networks for the split-bf16 kernels, made by the reference's modules


@pytest.mark.parametrize('name', sorted(CASES) + sorted(X3_NETS))
def test_torch_port_layers(name):
    c = CASES[name] if name in CASES else X3_NETS[name]
    _, specs = parse_vgsl(c['spec'])
    ref = CpuRecognizer(specs, c['sd'])
    if c['lens'] is None:
        y, _ = ref.forward(c['x'])
        np.testing.assert_allclose(y.numpy(), c['y'], atol=2e-6, rtol=1e-5)
        if 'olens_probe' in c:
            _, olens = ref.forward(c['x'], c['lens_probe'], reference_batched=True)
            assert olens.tolist() == c['olens_probe'].tolist()
    else:
        y, olens = ref.forward(_zero_pad_x(c['x'], c['lens']), c['lens'])
        for i, want in enumerate(c['ys']):
            w = want.shape[3]
            np.testing.assert_allclose(y[i:i + 1, ..., :w].numpy(), want, atol=LAYER_TOL, rtol=1e-4)


def _sha(t) -> str:
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def _bench_model(spec):
    import kraken_amd
    torch.manual_seed(0)
    return kraken_amd.TorchVGSLModel(vgsl=spec)


@pytest.mark.parametrize('spec,fixture', [(BENCH_A, 'bench_a.npz'), (BENCH_B, 'bench_b.npz')])
def test_oracles_on_bench_networks(spec, fixture):
    z = load_golden(fixture)
    m = _bench_model(spec)
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    _, specs = parse_vgsl(spec)
    ref = CpuRecognizer(specs, sd)
    x = synth_input(4, 400)
    keep = z['n4w400_keep'].tolist()
    # torch port: the reference's own operators -> (almost) bit-identical
    y, _ = ref.forward(x)
    np.testing.assert_allclose(y[keep].squeeze(2).numpy(), z['n4w400_logits'], atol=1e-5)
    probs = y.softmax(1).squeeze(2)
    dec = torch_greedy(probs, [probs.shape[2]] * 4)
    want = arr_to_tuples(z['n4w400_tuples'], z['n4w400_counts'])
    assert [[t[:3] for t in l] for l in dec] == [[t[:3] for t in l] for l in want]
    # numpy oracle on one line (slow python LSTM loop -> keep it small)
    yn, _ = np_oracle.forward(specs, sd, x[:1].numpy())
    np.testing.assert_allclose(yn[0, :, 0, :], z['n4w400_logits'][0], atol=2e-4)
    dn = np_oracle.greedy_decode(np_oracle.softmax_c(yn[:, :, 0, :]))
    assert [t[:3] for t in dn[0]] == [t[:3] for t in want[0]]
    for a, b in zip(dn[0], want[0]):
        assert abs(a[3] - b[3]) < 1e-4


def test_port_on_the_line_by_line_fixture_of_configs_2_and_4():
    """oracle/torch_port.py (what bench.py's cpu_baseline leg and its in-bench parity check run) against kraken's tuples and
    strings of bench_lines.npz: eight lines of the benchmark tensor, six ragged lines in one masked batch."""
    from kraken_amd.codec import PytorchCodec
    from tests.specs import bench_codec
    z = load_golden('bench_lines.npz')
    m = _bench_model(BENCH_A)
    assert json.loads(str(z['state_digest'])).keys() == m.state_dict().keys()
    _, specs = parse_vgsl(BENCH_A)
    ref = CpuRecognizer(specs, {k: v.numpy() for k, v in m.state_dict().items()})
    codec = PytorchCodec(bench_codec())
    want = arr_to_tuples(z['cfg2_tuples'], z['cfg2_counts'])
    strs = json.loads(str(z['cfg2_strings']))
    x = synth_input(256, 1200)[:8]
    got = ref.predict_labels(x, [1200] * 8)
    assert [[t[:3] for t in l] for l in got] == [[t[:3] for t in l] for l in want[:8]]
    assert [''.join(c for c, *_ in codec.decode(l)) for l in got] == strs[:8]
    want = arr_to_tuples(z['cfg4_tuples'], z['cfg4_counts'])
    strs = json.loads(str(z['cfg4_strings']))
    widths = z['cfg4_widths'].tolist()
    pick = [i for i in (0, 100, 333, 512, 800, 1023) if z['cfg4_margin'][i] > 1e-5]
    xb = torch.zeros(len(pick), 1, 48, max(widths[i] for i in pick))
    for k, i in enumerate(pick):
        xb[k, ..., :widths[i]] = synth_input(1, widths[i], seed=50000 + i)[0]
    got = ref.predict_labels(xb, [widths[i] for i in pick])
    assert [[t[:3] for t in l] for l in got] == [[t[:3] for t in want[i]] for i in pick]
    assert [''.join(c for c, *_ in codec.decode(l)) for l in got] == [strs[i] for i in pick]


def test_port_on_config3_shard_and_the_height_120_default_spec():
    """oracle/torch_port.py against kraken's tuples / strings of bench_lines_r6.npz: sixteen lines of config 3's 2048-line shard and
    eight lines (+ the logits of four) of kraken's default height-120 recogniser (kraken/configs/vgsl.py:102)."""
    from kraken_amd.codec import PytorchCodec
    from kraken_amd.specs import DEFAULT_H120
    from tests.specs import bench_codec
    z = load_golden('bench_lines_r6.npz')
    codec = PytorchCodec(bench_codec())
    for tag, spec, h, rows in (('cfg3', BENCH_A, 48, [5 * 16 + k for k in range(16)]), ('h120', DEFAULT_H120, 120, list(range(8)))):
        assert str(z[f'{tag}_spec']) == spec
        m = _bench_model(spec)
        if tag == 'h120':
            from tests.helpers import portable_weights
            portable_weights(m, seed=120)
        assert {k: _sha(v) for k, v in m.state_dict().items()} == json.loads(str(z[f'{tag}_state_digest']))
        _, specs = parse_vgsl(spec)
        ref = CpuRecognizer(specs, {k: v.numpy() for k, v in m.state_dict().items()})
        want = arr_to_tuples(z[f'{tag}_tuples'], z[f'{tag}_counts'])
        strs = json.loads(str(z[f'{tag}_strings']))
        x = synth_input(16, 1200, seed=30000 + 5, h=h) if tag == 'cfg3' else synth_input(64, 1200, seed=1200, h=h)[:8]
        keep = [k for k, i in enumerate(rows) if z[f'{tag}_margin'][i] > 1e-5]
        got = ref.predict_labels(x, [1200] * len(x))
        assert [[t[:3] for t in got[k]] for k in keep] == [[t[:3] for t in want[rows[k]]] for k in keep]
        assert [''.join(c for c, *_ in codec.decode(got[k])) for k in keep] == [strs[rows[k]] for k in keep]
        if tag == 'h120':
            logits, _ = ref.forward(x[:4], [1200] * 4)
            assert float((torch.as_tensor(logits) - torch.from_numpy(z['h120_logits4'])).abs().max()) < 2e-5


@pytest.mark.parametrize('tag', ['bidi1024', 'fwd1280', 'peep832', 'classic'])
def test_port_on_hidden_sizes_above_768(tag):
    """The oracle (torch port) against the reference's logits of big_lstm.npz -- the checker of the GPU test of the same name."""
    import kraken_amd
    from tests.helpers import portable_weights
    z = load_golden('big_lstm.npz')
    spec = str(z[f'{tag}_spec'])
    m = kraken_amd.TorchVGSLModel(vgsl=spec)
    portable_weights(m, seed=int(z[f'{tag}_seed']))
    assert {k: _sha(v) for k, v in m.state_dict().items()} == json.loads(str(z[f'{tag}_state_digest']))
    _, specs = parse_vgsl(spec)
    ref = CpuRecognizer(specs, {k: v.numpy() for k, v in m.state_dict().items()})
    lens = z[f'{tag}_lens'].tolist()
    y, olens = ref.forward(torch.from_numpy(z[f'{tag}_x']), lens if 'peep' not in tag else None)
    assert olens is None or list(olens) == z[f'{tag}_olens'].tolist()
    want = torch.from_numpy(z[f'{tag}_y'])
    for i, l in enumerate(z[f'{tag}_olens'].tolist()):        # ('classic': the spelled-out height collapse S1(1x12)1,3 of kraken's classic spec)
        assert float((torch.as_tensor(y)[i, ..., :l] - want[i, ..., :l]).abs().max()) < 1e-5


def test_oracles_ragged_equals_per_line_reference():
    """Masked padding == the reference's per-line (batch 1) result, for BENCH-A ragged widths."""
    z = load_golden('bench_a.npz')
    m = _bench_model(BENCH_A)
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    _, specs = parse_vgsl(BENCH_A)
    widths = z['ragged_widths'].tolist()
    x = synth_input(len(widths), 800, seed=4321)
    for i, w in enumerate(widths):
        x[i, ..., w:] = 0
    ref = CpuRecognizer(specs, sd)
    y, olens = ref.forward(x, widths)
    for i in range(len(widths)):
        want = z[f'ragged{i}_logits']
        assert olens[i] == want.shape[1]
        np.testing.assert_allclose(y[i, :, 0, :olens[i]].numpy(), want, atol=2e-5)
    # and the reference's own batched path is NOT batch invariant (documented in SURVEY 8a)
    assert z['ragged_batched_maxdiff'][1:].max() > 1e-3
    yb, _ = ref.forward(x, widths, reference_batched=True)
    diffs = [float(np.abs(yb[i, :, 0, :olens[i]].numpy() - z[f'ragged{i}_logits']).max()) for i in range(len(widths))]
    np.testing.assert_allclose(diffs, z['ragged_batched_maxdiff'], atol=1e-4)


def test_oracles_known_answer_overfit_model():
    """tests/test_rpred.py:352-358, :453-462 of the reference, through both oracles."""
    z = load_golden('overfit.npz')
    spec = str(z['spec'])
    sd = {k[3:]: z[k] for k in z.files if k.startswith('sd/')}
    _, specs = parse_vgsl(spec)
    meta = json.loads(str(z['meta']))
    l2c = {tuple(v): k for k, v in meta['codec'].items()}
    for pad in (1, 16):
        line = z[f'pad{pad}_line'][None]
        want_tuples = arr_to_tuples(z[f'pad{pad}_tuples'], z[f'pad{pad}_counts'])[0]
        y, _ = CpuRecognizer(specs, sd).forward(line)
        np.testing.assert_allclose(y.squeeze(2).numpy(), z[f'pad{pad}_logits'], atol=1e-5)
        yn, _ = np_oracle.forward(specs, sd, line)
        np.testing.assert_allclose(yn[:, :, 0, :], z[f'pad{pad}_logits'], atol=2e-4)
        probs = np_oracle.softmax_c(yn[:, :, 0, :])
        np.testing.assert_allclose(probs, z[f'pad{pad}_probs'], atol=1e-5)
        dec = np_oracle.greedy_decode(probs)[0]
        assert [t[:3] for t in dec] == [t[:3] for t in want_tuples]
        chars = np_oracle.codec_decode(l2c, dec)
        assert ''.join(c for c, *_ in chars) == str(z[f'pad{pad}_string_display'])
    # display-order string of the no-bidi known answer (tests/test_rpred.py:462)
    assert str(z['pad16_string_display']) == str(z['string_rpred_pad16_nobidi']) == 'ܕܗܣܐܕ ܪܝ .ܡܡ ܐܠܠ ܗܠ ܐܘܗ ܟܘܗܢ ܡܡ ܐܠ'
    assert str(z['string_rpred_pad1_bidi']) == 'ܡ ܘܡ ܗ ܡܕܐ ܐ ܐܐ ܡ ܗܗܐܐܐܕ'


def test_oracles_on_the_other_reference_recognisers():
    """overfit_newpoly.mlmodel, overfit_bl.safetensors, overfit_bl_newpoly.safetensors (SURVEY.md 8c list item 1):
    reference logits / tuples / strings of the fixture line through both transform branches."""
    z = load_golden('overfit_models.npz')
    for fi, fname in enumerate(json.loads(str(z['files']))):
        spec = str(z[f'm{fi}_spec'])
        sd = {k.split('_sd/')[1]: z[k] for k in z.files if k.startswith(f'm{fi}_sd/')}
        _, specs = parse_vgsl(spec)
        meta = json.loads(str(z[f'm{fi}_meta']))
        l2c = {tuple(v): k for k, v in meta['codec'].items()}
        for br in ('dewarp', 'resize'):
            tag = f'm{fi}_{br}'
            line = z[f'{tag}_line'][None]
            y, _ = CpuRecognizer(specs, sd).forward(line)
            np.testing.assert_allclose(y.squeeze(2).numpy(), z[f'{tag}_logits'], atol=2e-5, err_msg=fname)
            yn, _ = np_oracle.forward(specs, sd, line)
            np.testing.assert_allclose(yn[:, :, 0, :], z[f'{tag}_logits'], atol=5e-4, err_msg=fname)
            dec = np_oracle.greedy_decode(np_oracle.softmax_c(yn[:, :, 0, :]))[0]
            want = arr_to_tuples(z[f'{tag}_tuples'], z[f'{tag}_counts'])[0]
            assert [t[:3] for t in dec] == [t[:3] for t in want], fname
            assert ''.join(c for c, *_ in np_oracle.codec_decode(l2c, dec)) == str(z[f'{tag}_string'])


def test_np_greedy_decode_semantics():
    probs = np.array([[[0.1, 0.9, 0.9, 0.2, 0.2, 0.6],
                       [0.8, 0.05, 0.05, 0.7, 0.5, 0.3],
                       [0.1, 0.05, 0.05, 0.1, 0.3, 0.1]]], dtype=np.float32)   # (1, C=3, T=6)
    # labels per step: 1,0,0,1,1,0 -> runs of label 1 at [0,0] and [3,4]
    assert np_oracle.greedy_decode(probs) == [[(1, 0, 0, pytest.approx(0.8)), (1, 3, 4, pytest.approx(0.7))]]
    # ties resolve to the lowest class index
    tie = np.array([[[0.5], [0.5]]], dtype=np.float32)
    assert np_oracle.greedy_decode(tie) == [[]]
    with pytest.raises(ValueError):
        np_oracle.greedy_decode(np.zeros((2, 3, 4), np.float32))
    assert np_oracle.greedy_decode(probs, [3]) == [[(1, 0, 0, pytest.approx(0.8))]]


def test_dewarp_restatement_is_bit_exact_against_scipy_and_the_reference_fixtures():
    """
    oracle/np_oracle.py:center_normalize_np spells out the scipy.ndimage arithmetic of the CenterNormalizer dewarp (summation orders,
    boundary handling, integer truncation) -- the specification csrc/dewarp.hip follows.  Bit for bit against the scipy-based
    transform (kraken_amd/transforms.py:center_normalize, itself pinned to the reference's outputs in transforms.npz) on the
    fixture lines and on random lines of many heights, and through the float stage against the reference's tensors.
    """
    import json
    from kraken_amd.transforms import center_normalize, dewarp_tables
    from oracle.np_oracle import center_normalize_np, _gauss_weights
    z = load_golden('transforms.npz')
    for i, c in enumerate(json.loads(str(z['cases']))):
        if not c['valid_norm'] or c.get('channels', 1) != 1:
            continue
        arr = z[f'im{i}'].astype(np.float64)
        got = center_normalize_np(arr, c['height'])
        assert np.array_equal(got, center_normalize(arr, c['height']))
        t = 1.0 - np.clip(got, 0, 255).astype(np.uint8).astype(np.float32) / np.float32(255.0)      # array2pil truncation, / 255, invert
        pad = c['pad']
        inner = z[f'out{i}'][0][:, pad:z[f'out{i}'].shape[2] - pad] if pad else z[f'out{i}'][0]
        np.testing.assert_allclose(t, inner if pad else inner.max() - (inner.max() - inner), atol=1e-7) if pad else None
    rng = np.random.default_rng(3)
    for _ in range(10):
        h, w = int(rng.integers(20, 100)), int(rng.integers(30, 400))
        arr = np.full((h, w), 255.0)
        yc = (h / 2 + 0.2 * h * np.sin(np.arange(w) / rng.uniform(15, 60))).astype(int)
        for x in range(0, w, 2):
            if rng.random() < 0.7:
                arr[max(yc[x] - rng.integers(1, max(h // 3, 2)), 0):min(yc[x] + rng.integers(1, max(h // 3, 2)), h), x:x + 2] = rng.integers(0, 120)
        assert np.array_equal(center_normalize_np(arr, 48), center_normalize(arr, 48))
    # the 60 lines of the GPU record test (tests/test_gpu_parity.py): lines 3, 22, 50, 53 of this set are the ones on which a device
    # build with fused multiply-adds moved a few columns' centre by a row -- the restatement itself is scipy's on all of them
    from tests.helpers import wavy_line as _wavy_line
    rs = np.random.RandomState(9)
    for i in range(60):
        h, w = int(rs.randint(30, 90)), int(rs.randint(200, 1000))
        arr = _wavy_line(rs, h, w).astype(np.float64)
        assert np.array_equal(center_normalize_np(arr, 48), center_normalize(arr, 48)), i
    # the weight tables the device receives are the restatement's (= scipy's _gaussian_kernel1d)
    tab, index = dewarp_tables([37, 60])
    off, r0, r1, r2 = index[60]
    w1, rr = _gauss_weights(60 * 1.0)
    assert rr == r1 and np.array_equal(tab[off + 2 * r0 + 1:off + 2 * r0 + 1 + 2 * r1 + 1], w1)


def _reshape_cases():
    z = load_golden('reshape_random.npz')
    return z, json.loads(str(z['cases']))


def test_random_reshapes_and_additions_like_the_reference():
    """240 random `S…` / `A…` layers made by the reference (tests/golden/make_golden.py: reshape_random_fixture): every split axis and
    target, `-1` parts, parts that do not divide, fixed and variable widths, the batch axis.  The parser refuses what the reference
    refuses with the same exception type (and, earlier than the reference, an Addition whose chunk exceeds a fixed axis: there the
    reference's forward raises), derives the same static output shape, and both oracles move an arange tensor to exactly the same
    places and return the same seq_lens."""
    z, cases = _reshape_cases()
    ran = 0
    for c in cases:
        try:
            _, specs = parse_vgsl(c['spec'])
            err = None
        except Exception as e:      # noqa: BLE001
            specs, err = None, type(e).__name__
        if not c['ok']:
            assert err == c['error'], (c['spec'], err, c['error'])
            continue
        if specs is None:
            assert err == 'ValueError' and ' A' in c['spec'] and not c.get('runs'), (c['spec'], err)
            continue
        assert list(specs[-1].out_shape) == c['static'], (c['spec'], specs[-1].out_shape, c['static'])
        if not c.get('runs'):
            continue
        n, ch, h, w = c['n'], specs[0].in_shape[1], specs[0].in_shape[2], c['w']
        x = np.arange(n * ch * h * w, dtype=np.float32).reshape(n, ch, h, w)
        want = z[f"y{c['i']}"]
        y, _ = np_oracle.forward(specs, {}, x)
        assert y.shape == want.shape and np.array_equal(y.astype(np.int32), want), c['spec']
        yt, _ = CpuRecognizer(specs, {}).forward(x)
        assert np.array_equal(yt.numpy().astype(np.int32), want), c['spec']
        _, ol = np_oracle.forward(specs, {}, x, c['lens'])
        assert list(ol) == c['olens'], (c['spec'], c['lens'], ol, c['olens'])
        _, olt = CpuRecognizer(specs, {}).forward(x, c['lens'], reference_batched=True)
        assert olt.tolist() == c['olens'], c['spec']
        ran += 1
    assert ran >= 150
