"""Dev tool (GPU): the colour recogniser (BENCH-A with three input channels) alone, per launch group, with its first convolution on the
three-channel first-layer kernel (conv1_x3.hip, round 5) and on the exact-f32 kernel it replaced (KRK_NO_CONV1_X3, read when the plan
is built); the logits of both against each other and -- 8 lines -- against the CPU oracle.   python tools/rgb_first_layer.py [N] [W]"""
import os
import sys

sys.path.insert(0, '.')
import torch  # noqa: E402

import kraken_amd  # noqa: E402
from kraken_amd.engine import RecognitionEngine  # noqa: E402
from kraken_amd.specs import BENCH_A_RGB, bench_codec  # noqa: E402
from oracle.torch_port import CpuRecognizer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
x = torch.rand(N, 3, 48, W, generator=torch.Generator().manual_seed(1)).cuda()
outs = {}
for label, env in (('conv1_x3 (3 channels)', None), ('exact-f32 first layer', 'KRK_NO_CONV1_X3')):
    if env:
        os.environ[env] = '1'
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=BENCH_A_RGB, codec=bench_codec()).to('cuda')
    m.nn.set_precision('bf16x3')
    eng = RecognitionEngine(m, device=0, max_batch=N, max_width=W, slots=1)
    eng.set_profiling(True)
    best = None
    for r in range(6):
        eng.submit(x)
        eng.collect()
        t = [(n_, ms) for n_, ms, _ in eng.layer_times()[0]]
        if r:
            best = t if best is None else [(a[0], min(a[1], b[1])) for a, b in zip(best, t)]
    y, _ = m.nn(x)
    outs[label] = y.float().cpu()
    eng.close()
    if env:
        os.environ.pop(env)
    print(f'{label}: total {sum(ms for _, ms in best):.3f} ms |', ' '.join('%s=%.3f' % (n_, ms) for n_, ms in best), flush=True)
a, b = outs.values()
print('max |d logit| between the two first layers: %.2e' % (a - b).abs().max().item())
ref = CpuRecognizer(m.layer_specs, {k: v.cpu() for k, v in m.state_dict().items()})
want, _ = ref.forward(x[:8].cpu())
print('max |d logit| against the CPU oracle (8 lines): %.2e (new), %.2e (exact-f32 first layer)' %
      ((a[:8] - want).abs().max().item(), (b[:8] - want).abs().max().item()))
