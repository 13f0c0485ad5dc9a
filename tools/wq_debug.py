"""Dev probe of lstm_wq.hip: one recurrent layer, the cluster kernel of rounds 2-4 (KRK_LSTM_V=3) against the new one (4), several
forwards on the same plan.   python tools/wq_debug.py N T [repeats]"""
import sys, os, torch
sys.path.insert(0, '.')
import kraken_amd
torch.manual_seed(0)
N, T = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
m = kraken_amd.TorchVGSLModel(vgsl=os.environ.get('WQ_SPEC', '[1,1,0,400 Lbx200]'))
m.nn.set_precision('bf16x3'); m.to('cuda')
x = torch.rand(N, 400, 1, T, device='cuda')
os.environ['KRK_LSTM_V'] = '3'
y3, _ = m.nn(x); y3 = y3.clone()
os.environ['KRK_LSTM_V'] = '1'
y1, _ = m.nn(x); y1 = y1.clone()
os.environ['KRK_LSTM_V'] = '4'
for r in range(reps):
    try:
        y4, _ = m.nn(x)
        print('N', N, 'T', T, 'rep', r, 'max |wq - ws|', (y4 - y3).abs().max().item(), 'identical', torch.equal(y4, y3), 'equals the streaming kernel', torch.equal(y4, y1), flush=True)
    except Exception as e:
        print('ERR', str(e)[:120], flush=True)
if not torch.equal(y4, y3):
    d = (y4 - y3).abs()
    bad = (d > 0).any(dim=1).squeeze(1) if d.dim() == 4 else (d > 0)
    bad = bad.reshape(N, -1)
    lines = bad.any(dim=1).nonzero().flatten().tolist()
    times = bad.any(dim=0).nonzero().flatten().tolist()
    print('lines that differ', len(lines), lines[:40])
    print('times that differ', len(times), times[:40], '...', times[-5:])
