import sys, torch
sys.path.insert(0, '.')
import kraken_amd
from oracle.torch_port import CpuRecognizer
from kraken_amd.specs import BENCH_A
specs = [('A', BENCH_A, 4, 400, [400, 307, 201, 399], 48, 1),
         ('small', '[1,8,0,1 Cr3,3,16 Mp2,2 Cr3,5,32 S1(1x0)1,3 Lbx8 O1c12]', 3, 90, [90, 61, 17], 8, 1),
         ('nopool-tanh', '[1,6,0,3 Ct3,3,16 Cr3,7,48,1,2 Cl1,1,32 S1(1x0)1,3 Lfx16 Lbx8 O1c7]', 2, 77, None, 6, 3)]
for name, spec, N, W, lens, H, C in specs:
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=spec)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(N, C, H, W, generator=g)
    if lens:
        for i, l in enumerate(lens): x[i, ..., l:] = 0
    ref = CpuRecognizer(m.layer_specs, m.state_dict())
    want, wl = ref.forward(x, lens)
    m.to('cuda')
    out = {}
    for prec in ('f32', 'bf16x3'):
        m.nn.set_precision(prec)
        got, gl = m.nn(x.cuda(), None if lens is None else torch.tensor(lens))
        got = got.cpu()
        if lens is None: d = (got - want).abs().max().item()
        else: d = max((got[i, ..., :wl[i]] - want[i, ..., :wl[i]]).abs().max().item() for i in range(N))
        lab = (got.squeeze(2).argmax(1) != want.squeeze(2).argmax(1)).sum().item()
        print(name, prec, 'max|dlogit|=%.3e' % d, 'label mismatches', lab, flush=True)
