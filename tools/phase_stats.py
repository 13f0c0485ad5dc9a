"""Dev tool (GPU, -DKRK_ABLATE build): where the waves of conv_x3p.hip spend their cycles on BENCH-A (one batch in
flight).  Probe bit 64 makes every wave sum the shader cycles of its phases (common.h KRK_PHASES); printed per kernel as the mean
cycles per wave and the share of the wave's lifetime.
    python -m kraken_amd.build --ablate && python tools/phase_stats.py [NAME=VALUE,...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, '.')
os.environ['KRAKEN_AMD_LIB'] = os.path.abspath('kraken_amd/libkraken_amd_ablate.so')
for kv in sys.argv[1:]:
    for item in kv.split(','):
        k, v = item.split('=')
        os.environ[k] = v
os.environ['KRK_X3_DBG'] = str(int(os.environ.get('KRK_X3_DBG', '0')) | 64)
import torch  # noqa: E402

import kraken_amd  # noqa: E402
from kraken_amd import _lib  # noqa: E402
from kraken_amd.specs import BENCH_A, bench_codec  # noqa: E402

torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=BENCH_A, codec=bench_codec()).to('cuda')
m.nn.set_precision('bf16x3')
x = torch.rand(256, 1, 48, 1200, generator=torch.Generator().manual_seed(1)).cuda()
lib = _lib.load()
buf = (C.c_ulonglong * 8)()
m.nn(x)
for which in (0,):
    lib.krk_debug_phase_stats(which, buf, 1)
reps = 3
for _ in range(reps):
    m.nn(x)
names = {0: ('conv_x3p (both launches)', ['prologue', 'copy waits', 'barrier', 'copy issue', 'reads+MFMA', 'epilogue'])}
for which in (0,):
    n = lib.krk_debug_phase_stats(which, buf, 1)
    title, ph = names[which]
    waves = buf[len(ph)]
    if n <= 0 or not waves:
        print(title, 'no data')
        continue
    tot = sum(buf[i] for i in range(len(ph)))
    print(f'{title}: {waves // reps} waves per forward, mean {tot / waves:.0f} cycles per wave |',
          ' '.join(f'{p}={buf[i] / waves:.0f} ({100.0 * buf[i] / tot:.0f}%)' for i, p in enumerate(ph)), flush=True)
