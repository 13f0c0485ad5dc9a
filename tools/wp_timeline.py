"""Dev tool (GPU, -DKRK_ABLATE build): time line of the stages 64..95 of cluster 0 / slice 0 of the LAST lstm_wp launch.
    python -m kraken_amd.build --ablate && KRAKEN_AMD_LIB=kraken_amd/libkraken_amd_ablate.so python tools/wp_timeline.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, '.')
os.environ.setdefault('KRAKEN_AMD_LIB', os.path.abspath('kraken_amd/libkraken_amd_ablate.so'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import kraken_amd  # noqa: E402
from kraken_amd import _lib  # noqa: E402
from kraken_amd.specs import BENCH_A  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=BENCH_A)
m.nn.set_precision('bf16x3')
m.to('cuda')
x = torch.rand(N, 1, 48, 1200).cuda()
for _ in range(3):
    m.nn(x)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 1024)()
rc = lib.krk_debug_wp_timeline(buf, 1024)
t = np.array(buf[:], dtype=np.int64).reshape(32, 32)
print('rc', rc, '  columns: compute wave 0: barrier exit | operands issued | MFMAs done | next barrier entry ;; gather wave 8: barrier exit | first poll back | tags ok | next barrier entry')
base = t[0, 0]
for k in range(32):
    r = t[k]
    print(f'stage {64 + k:3d}: C +{r[0] - base:6d} ops {r[1] - r[0]:5d} mfma {r[2] - r[0]:5d} bar-in {r[3] - base:6d} | '
          f'G +{r[4] - base:6d} poll1 {r[5] - r[4]:5d} ok {r[6] - r[4]:5d} bar-in {r[7] - base:6d}')
d = np.diff(t[:, 0])
print('stage period (cycles of s_memtime): mean', d.mean(), 'min', d.min(), 'max', d.max())
print('barrier entry of every wave relative to the barrier exit of the stage of wave 0 (compute 0..7 | gather 8..11), and the last to arrive:')
for k in range(1, 32):
    ent = t[k, 8:20] - t[k - 1, 20]            # entry into the barrier that ENDS stage k-1 ... measured from that stage's start
    print(f'stage {63 + k:3d}: ' + ' '.join('%5d' % v for v in ent[:8]) + ' | ' + ' '.join('%5d' % v for v in ent[8:]) + f'   last: wave {int(np.argmax(ent))}  period {t[k, 20] - t[k - 1, 20]}')
if int(os.environ.get('KRK_LSTM_DBG', '0')) & 8192:
    print('wave 0: cycles after its barrier exit at which K block 0..6 was done (MFMAs issued + that block\'s share of the gate math)')
    for k in range(8, 32):
        print(f'stage {64 + k:3d}: ' + ' '.join('%5d' % (t[k, 9 + j] - t[k, 20]) for j in range(7)))
