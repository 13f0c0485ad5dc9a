"""BASELINE.json config 5: forward of the BLLA baseline segmenter (reference default spec, kraken/configs/vgsl.py:122 plus
an 8-class O2l heatmap head) on a 4k x 3k page scaled to the network height of 1800 -> (1, 3, 1800, 1350).
Times the HIP forward and checks it against the CPU oracle (oracle/torch_port.py).  Usage: python tools/blla_forward.py [H W]"""
import sys
import time

import torch

sys.path.insert(0, '.')
import kraken_amd  # noqa: E402
from oracle.torch_port import CpuRecognizer  # noqa: E402

BLLA = ('[1,1800,0,3 Cr7,7,64,2,2 Gn32 Cr3,3,128,2,2 Gn32 Cr3,3,128 Gn32 Cr3,3,256 Gn32 Cr3,3,256 Gn32 '
        'Lbx32 Lby32 Cr1,1,32 Gn32 Lby32 Lbx32 O2l8]')
_pos = [a for a in sys.argv[1:] if not a.startswith('--')]
H, W = (int(_pos[0]), int(_pos[1])) if len(_pos) > 1 else (1800, 1350)
spec = BLLA.replace('[1,1800,0,3', f'[1,{H},0,3')
torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=spec)
x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(1))
m.to('cuda')
if '--x3' in sys.argv:
    m.nn.set_precision('bf16x3')   # convolutions + GroupNorm on the bf16 cores, 2-D LSTMs and the tail on the f32 kernels
xd = x.cuda()
y, _ = m.nn(xd)
torch.cuda.synchronize()
a, b = torch.cuda.Event(True), torch.cuda.Event(True)
a.record()
for _ in range(5):
    y, _ = m.nn(xd)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f'HIP forward {tuple(x.shape)} -> {tuple(y.shape)}: {ms:.2f} ms per page', flush=True)
t = time.time()
want, _ = CpuRecognizer(m.layer_specs, {k: v.cpu() for k, v in m.state_dict().items()}).forward(x)
print(f'CPU oracle {time.time() - t:.1f} s on {torch.get_num_threads()} threads; max|d logit| = {(y.cpu() - want).abs().max().item():.3e} '
      f'(|logit| max {want.abs().max().item():.2f})', flush=True)
if '--layers' in sys.argv:
    import ctypes as C
    from kraken_amd import _lib
    plan = m.nn.plan(xd.device.index or 0)
    lib = _lib.load()
    _lib.check(lib.krk_plan_set_profiling(plan.handle, 1))
    m.nn(xd)
    torch.cuda.synchronize()
    n = lib.krk_plan_num_steps(plan.handle)
    msv = (C.c_float * n)()
    _lib.check(min(lib.krk_plan_layer_ms(plan.handle, msv, n), 0))
    for i in range(n):
        print('  %-10s %7.3f ms' % (lib.krk_plan_layer_name(plan.handle, i).decode(), msv[i]))
