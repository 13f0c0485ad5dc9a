import sys, time, torch
sys.path.insert(0, '.')
import kraken_amd
from oracle.torch_port import CpuRecognizer
from kraken_amd.specs import BENCH_A, BENCH_B
for name, spec, N, W, lens in [('A-eq', BENCH_A, 3, 400, None), ('A-ragged', BENCH_A, 5, 400, [400, 307, 201, 399, 202]),
                               ('B-eq', BENCH_B, 3, 200, None), ('B-ragged', BENCH_B, 4, 200, [200,151,99,77]),
                               ('A-16', BENCH_A, 16, 800, None)]:
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=spec)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(N, 1, 48, W, generator=g)
    if lens is not None:
        for i, l in enumerate(lens): x[i, ..., l:] = 0
    ref = CpuRecognizer(m.layer_specs, m.state_dict())
    t=time.time(); want, wl = ref.forward(x, lens); tc=time.time()-t
    m.to('cuda')
    t=time.time(); got, gl = m.nn(x.cuda(), None if lens is None else torch.tensor(lens)); torch.cuda.synchronize(); tg=time.time()-t
    got = got.cpu()
    print(name, tuple(got.shape), tuple(want.shape), 'olens', gl, wl)
    if lens is None:
        d = (got-want).abs().max().item()
    else:
        d = max((got[i,...,:wl[i]]-want[i,...,:wl[i]]).abs().max().item() for i in range(N))
    lab_g = got.squeeze(2).argmax(1); lab_w = want.squeeze(2).argmax(1)
    print(f'  max|dlogit|={d:.3e} label mismatches={(lab_g!=lab_w).sum().item()} cpu={tc:.3f}s gpu={tg:.3f}s', flush=True)
    rec_g = m.nn.recognize(x.cuda(), None if lens is None else torch.tensor(lens))[0].tuples()
    rec_w = ref.predict_labels(x, lens)
    same = all([(a[0],a[1],a[2]) for a in p]==[(b[0],b[1],b[2]) for b in q] for p,q in zip(rec_g, rec_w))
    cd = max([abs(a[3]-b[3]) for p,q in zip(rec_g, rec_w) for a,b in zip(p,q)] or [0])
    print('  tuples identical:', same, 'n tuples', sum(map(len,rec_w)), 'max conf diff', cd, flush=True)
