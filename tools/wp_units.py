"""Dev tool (GPU): which (line, unit, direction) of the recurrent layer's output differ between KRK_LSTM_V=4 and =1?
The linear layer behind the LSTM is set to a selection matrix, so the logits ARE h."""
import os
import subprocess
import sys

sys.path.insert(0, '.')


def child(v, T, N, out, H=200):
    import numpy as np
    import torch
    import kraken_amd
    spec = '[1,48,0,1 Cr3,13,32 Mp2,2 Cr3,13,32 Mp2,2 Cr3,9,64 Mp2,2 Cr3,9,64 S1(1x0)1,3 Lbx%d O1c%d]' % (H, 2 * H)
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=spec)
    with torch.no_grad():
        lin = [mod for mod in m.nn.children() if hasattr(mod, 'lin')][0].lin
        lin.weight.copy_(torch.eye(2 * H))
        lin.bias.zero_()
    m.nn.set_precision('bf16x3')
    m.to('cuda')
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, 1, 48, T * 8, generator=g).cuda()
    y, _ = m.nn(x)
    torch.cuda.synchronize()
    np.save(out, y.float().cpu().numpy())


if __name__ == '__main__':
    if sys.argv[1] == 'child':
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]))
        sys.exit(0)
    import numpy as np
    T, N = int(sys.argv[1]), int(sys.argv[2])
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    ys = {}
    for v in (1, 4):
        out = f'/tmp/wpu_{v}.npy'
        r = subprocess.run([sys.executable, __file__, 'child', str(v), str(T), str(N), out, str(H)], env=dict(os.environ, KRK_LSTM_V=str(v)), capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-400:])
        ys[v] = np.load(out)[:, :, 0, :]          # (N, 2H, T)
    d = np.abs(ys[4] - ys[1])
    d = np.where(np.isnan(d), 1e9, d)
    bad = d > 1e-3
    print('bad entries', int(bad.sum()), 'of', bad.size)
    for n in range(N):
        if bad[n].any():
            us = np.nonzero(bad[n].any(axis=1))[0]
            ts = np.nonzero(bad[n].any(axis=0))[0]
            print(f'line {n:3d}: bad units {len(us)} (dir0 {int((us < H).sum())}, dir1 {int((us >= H).sum())}) first {us[:12].tolist()} last {us[-4:].tolist()} | t {ts[:8].tolist()}'
                  f' | wp {ys[4][n, us[0], ts[0]]:.4g} ref {ys[1][n, us[0], ts[0]]:.4g}')
