"""Probe of the recurrent kernels (dev tool, GPU): per-launch time of the streaming (v1) and the weight-stationary cluster (ws)
kernel on BENCH-A + agreement of their network outputs.

    python tools/lstm_v2_probe.py            # table of variants x batch sizes
    python tools/lstm_v2_probe.py --ablate   # phase prices from the -DKRK_ABLATE build (python -m kraken_amd.build --ablate)
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, '.')
from kraken_amd.specs import BENCH_A as SPEC  # noqa: E402


def child(N, T, dump):
    import ctypes as C
    import numpy as np
    import torch
    import kraken_amd
    from kraken_amd import _lib
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=SPEC)
    m.nn.set_precision('bf16x3')
    m.to('cuda')
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, 1, 48, T * 8, generator=g).cuda()
    lens = None
    if os.environ.get('PROBE_RAGGED'):
        lens = torch.tensor([8 * max(1, T - (7 * i) % (T // 2)) for i in range(N)])
    plan = m.nn.plan(0)
    lib = _lib.load()
    lib.krk_plan_set_profiling(plan.handle, 1)
    best = {}
    for _ in range(6):
        y, _ = m.nn(x, lens)
        torch.cuda.synchronize()
        n = lib.krk_plan_num_steps(plan.handle)
        ms = (C.c_float * n)()
        lib.krk_plan_layer_ms(plan.handle, ms, n)
        for i in range(n):
            k = lib.krk_plan_layer_name(plan.handle, i).decode() + f'@{i}'
            best[k] = min(best.get(k, 1e9), ms[i])
    if dump:
        np.save(dump, y.float().cpu().numpy())
    out = {}
    for k, v in best.items():
        out.setdefault(k.split('@')[0], []).append(round(v, 4))
    print(json.dumps(out))


def run(env, N, T, dump=None):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    try:
        out = subprocess.run([sys.executable, __file__, 'child', str(N), str(T), dump or ''], env=e, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        return {'error': 'timeout'}
    try:
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        return {'error': (out.stderr or out.stdout)[-400:]}


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] or None)
        sys.exit(0)
    import numpy as np
    os.makedirs('gpurun_out', exist_ok=True)
    variants = [('v1', dict(KRK_LSTM_V=1)), ('ws g2', dict(KRK_LSTM_V=3, KRK_LSTM_G=2)), ('ws g4', dict(KRK_LSTM_V=3, KRK_LSTM_G=4))]
    if '--ablate' in sys.argv:
        lib = os.path.abspath('kraken_amd/libkraken_amd_ablate.so')
        for name, env in variants[1:2]:
            for dbg in (0, 1, 2, 4, 8, 16, 32, 3, 7, 23, 55):
                r = run(dict(env, KRAKEN_AMD_LIB=lib, KRK_LSTM_DBG=dbg), 256, 150)
                print('ablate', name, 'dbg', dbg, r.get('lstm_rec_x3', r), flush=True)
        sys.exit(0)
    if '--quick' in sys.argv:            # correctness at three sizes + the phase prices of the default (g2) variant
        lib = os.path.abspath('kraken_amd/libkraken_amd_ablate.so')
        for dbg in (0, 64, 128, 4, 16, 32, 1):     # 1 no exchange, 4 no MFMA, 16 no output pass, 32 no barrier; gather asked at slot start (64) / after the last block (128)
            r = run(dict(variants[1][1], KRAKEN_AMD_LIB=lib, KRK_LSTM_DBG=dbg), 256, 150)
            print('ablate ws g2 dbg', dbg, r.get('lstm_rec_x3', r), flush=True)
        variants = variants[:2]
        sizes = ((256, 150), (40, 60), (7, 33))
    else:
        variants = variants[:2]        # (the 4-groups-per-cluster variant is probed by tests/test_gpu_parity.py)
        sizes = ((256, 150), (40, 60), (7, 33), (1024, 150), (100, 300))
    for N, T in sizes:
        ref = None
        for name, env in variants:
            for ragged in (0, 1):
                dump = f'/tmp/probe_{name.replace(" ", "_")}_{N}_{ragged}.npy'
                e = dict(env)
                if ragged:
                    e['PROBE_RAGGED'] = 1
                r = run(e, N, T, dump)
                d = None
                if os.path.exists(dump):
                    y = np.load(dump)
                    if name == 'v1':
                        ref = ref or {}
                        ref[ragged] = y
                    elif ref and ragged in ref:
                        d = float(np.abs(y - ref[ragged]).max())
                        if name == 'ws g2':
                            ref['ws', ragged] = y
                        elif ('ws', ragged) in ref:
                            d = (d, 'bit-identical to ws' if np.array_equal(y, ref['ws', ragged]) else f'max |d| vs ws {float(np.abs(y - ref["ws", ragged]).max()):.3g}')
                print(f'N={N} T={T} ragged={ragged} {name:12s} rec={r.get("lstm_rec_x3", r)} xproj={r.get("lstm_xproj_x3")} maxdiff_vs_v1={d}', flush=True)
