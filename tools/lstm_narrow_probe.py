"""Which recurrent kernel for NARROW layers (H <= 128)?  Round 3 sent them to a third kernel (lstm_wp.hip) because lstm_ws.hip timed out
there; the cause is fixed, so the routing is a speed question again.  Per-launch time of the recurrent layers of
a two-layer network and exchange timeouts over the runs, per kernel (KRK_LSTM_V).  The round-4 run of this probe
(profiles/r04_lstm_narrow_probe.txt, made while lstm_wp.hip still existed: V=4) decided the routing and the removal of that kernel.
    python tools/lstm_narrow_probe.py            (GPU)"""
import ctypes as C
import json
import os
import subprocess
import sys

sys.path.insert(0, '.')


def child(H, N, T):
    import torch
    import kraken_amd
    from kraken_amd import _lib
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=f'[1,1,0,64 Lbx{H} Lbx{H} O1c40]')
    m.nn.set_precision('bf16x3')
    m.to('cuda')
    x = torch.rand(N, 64, 1, T).cuda()
    plan = m.nn.plan(0)
    lib = _lib.load()
    lib.krk_plan_set_profiling(plan.handle, 1)
    best, bad = {}, 0
    for _ in range(int(os.environ.get('PROBE_RUNS', '30'))):
        try:
            m.nn(x)
        except Exception:
            bad += 1
            continue
        n = lib.krk_plan_num_steps(plan.handle)
        ms = (C.c_float * n)()
        k = lib.krk_plan_layer_ms(plan.handle, ms, n)
        for i in range(k):
            name = (lib.krk_plan_layer_name(plan.handle, i) or b'?').decode() + str(i)
            best[name] = min(best.get(name, 1e9), ms[i])
    print(json.dumps({'rec': {k: round(v, 4) for k, v in best.items() if 'rec' in k}, 'timeouts': bad}))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(*(int(v) for v in sys.argv[2:5]))
        sys.exit(0)
    for N, T in ((256, 150), (32, 150)):
        for H in (16, 32, 64, 96, 128):
            for v, name in (('3', 'lstm_ws'), ('1', 'streaming')):
                env = dict(os.environ, KRK_LSTM_V=v)
                out = subprocess.run([sys.executable, __file__, 'child', str(H), str(N), str(T)], env=env, capture_output=True, text=True)
                line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
                print(f'N={N} T={T} H={H:3d} {name:10s} {line}', flush=True)
