"""Dev tool (GPU, -DKRK_ABLATE build): the BENCH-A convolution / projection kernels alone (one batch in flight) with phases switched off.
Bits (KRK_X3_DBG): conv_x3 1 no MFMA, 2 no input staging loads, 4 no stores, 8 no weight copies; conv_taps 1 no MFMA, 2 return after the
K loop, 4 no staging loads, 16 no stores; conv1 1 no MFMA, 2 no loads, 4 no stores; gemm_x3 1 no MFMA, 2 no copies, 4 no stores, 8 no LDS reads.
    python -m kraken_amd.build --ablate && python tools/x3_ablate.py"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, '.')
    import torch
    import kraken_amd
    from kraken_amd.engine import RecognitionEngine
    from kraken_amd.specs import BENCH_A, bench_codec
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=BENCH_A, codec=bench_codec()).to('cuda')
    m.nn.set_precision('bf16x3')
    x = torch.rand(256, 1, 48, 1200, generator=torch.Generator().manual_seed(1)).cuda()
    eng = RecognitionEngine(m, device=0, max_batch=256, max_width=1200, slots=1)
    eng.set_profiling(True)
    best = None
    for _ in range(5):
        eng.submit(x)
        try:
            eng.collect()
        except Exception:
            pass
        t = [(n_, ms) for n_, ms, _ in eng.layer_times()[0]]
        best = t if best is None else [(a[0], min(a[1], b[1])) for a, b in zip(best, t)]
    print(' '.join('%s=%.3f' % (n_, ms) for n_, ms in best if not n_.startswith('lstm_rec')), flush=True)
    sys.exit(0)

lib = os.path.abspath('kraken_amd/libkraken_amd_ablate.so')
for dbg in (0, 1, 2, 4, 8, 16, 6, 20, 7, 31):
    env = dict(os.environ, KRAKEN_AMD_LIB=lib, KRK_X3_DBG=str(dbg))
    out = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if 'conv' in l]
    print(f'KRK_X3_DBG={dbg:2d}:', line[-1] if line else (out.stderr or out.stdout)[-300:], flush=True)
