"""Ablation probe of the LSTM recurrent kernel (dev tool): per-phase cost via KRK_LSTM_DBG bits."""
import os, sys, subprocess, json
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch, kraken_amd, ctypes as C
    from kraken_amd import _lib
    spec = sys.argv[2]; N = int(sys.argv[3]); T = int(sys.argv[4])
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=spec).to('cuda')
    c = m.input[1]
    x = torch.rand(N, c, 1, T, device='cuda')
    plan = m.nn.plan(0)
    lib = _lib.load()
    lib.krk_plan_set_profiling(plan.handle, 1)
    for _ in range(3):
        m.nn(x)
    torch.cuda.synchronize()
    n = lib.krk_plan_num_steps(plan.handle)
    ms = (C.c_float * n)()
    lib.krk_plan_layer_ms(plan.handle, ms, n)
    print(json.dumps({lib.krk_plan_layer_name(plan.handle, i).decode() + str(i): round(ms[i], 3) for i in range(n)}))
    sys.exit(0)
N, T = 256, 150
for name, spec in [('xproj-only(linear 400->1600)', '[1,1,0,400 O1c1600]'), ('lstm', '[1,1,0,400 Lbx200]')]:
    for M in ('16', '32'):
        for dbg in ([0] if 'linear' in name else [0, 1, 2, 4, 8, 3, 15]):
            env = dict(os.environ, KRK_LSTM_DBG=str(dbg), KRK_LSTM_M=M)
            out = subprocess.run([sys.executable, __file__, 'child', spec, str(N), str(T)], env=env, capture_output=True, text=True)
            print(name, 'M', M, 'dbg', dbg, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
        if 'linear' in name: break
