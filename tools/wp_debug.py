"""Dev tool (GPU): where does the pipelined cluster kernel (KRK_LSTM_V=4) differ from the streaming kernel (V=1)?
Per case: NaN count, lines that differ, first differing time step.   python tools/wp_debug.py"""
import json
import os
import subprocess
import sys

sys.path.insert(0, '.')


def child(spec, N, T, scale):
    import numpy as np
    import torch
    import kraken_amd
    torch.manual_seed(0)
    m = kraken_amd.TorchVGSLModel(vgsl=spec)
    m.nn.set_precision('bf16x3')
    m.to('cuda')
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, 1, int(spec.split(',')[1]), T * scale, generator=g).cuda()
    try:
        y, _ = m.nn(x)
        torch.cuda.synchronize()
        np.save(os.environ['OUT'], y.float().cpu().numpy())
        print(json.dumps({'ok': True}))
    except Exception as e:
        print(json.dumps({'ok': False, 'err': str(e)[-160:]}))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    import numpy as np
    from kraken_amd.specs import BENCH_A
    SMALL = '[1,16,0,1 Cr5,7,16 Mp2,2 Cr3,12,32 Mp2,2 Cr3,3,32 S1(1x0)1,3 Lbx8 O1c7]'
    ONE = '[1,48,0,1 Cr3,13,32 Mp2,2 Cr3,13,32 Mp2,2 Cr3,9,64 Mp2,2 Cr3,9,64 S1(1x0)1,3 Lbx200 O1c20]'   # ONE recurrent layer
    ONEF = ONE.replace('Lbx200', 'Lfx200')
    SMALLF = SMALL.replace('Lbx8', 'Lfx8')
    cases = [('one', ONE, 64, 1, 8), ('one', ONE, 64, 2, 8), ('one', ONE, 64, 5, 8), ('small', SMALL, 64, 3, 4)]
    if '--full' in sys.argv:
        cases = [('A', BENCH_A, 40, 150, 8), ('A', BENCH_A, 64, 60, 8), ('A', BENCH_A, 128, 60, 8), ('A', BENCH_A, 64, 150, 8),
                 ('one', ONE, 64, 150, 8), ('one', ONE, 128, 40, 8), ('small', SMALL, 3, 75, 4), ('small', SMALL, 3, 45, 4), ('small', SMALL, 70, 75, 4)]
    libs = [('rel', {})]
    abl = os.path.abspath('kraken_amd/libkraken_amd_ablate.so')
    if os.path.exists(abl):
        libs.append(('direct-x', {'KRAKEN_AMD_LIB': abl, 'KRK_LSTM_DBG': '2'}))
    for name, spec, N, T, scale in cases:
        outs = {}
        for v in (1, 4):
            for lname, lenv in libs:
                out = f'/tmp/wpdbg_{v}_{lname}.npy'
                if os.path.exists(out):
                    os.remove(out)
                env = dict(os.environ, KRK_LSTM_V=str(v), OUT=out, **lenv)
                try:
                    r = subprocess.run([sys.executable, __file__, 'child', spec, str(N), str(T), str(scale)], env=env, capture_output=True, text=True, timeout=120)
                    msg = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]
                except subprocess.TimeoutExpired:
                    msg = 'TIMEOUT'
                outs[(v, lname)] = (np.load(out) if os.path.exists(out) else None, msg)
        for (v, lname), (y, msg) in outs.items():
            if v == 1:
                continue
            ref = outs[(1, lname)][0]
            if y is None or ref is None:
                print(f'{name} N={N} T={T} [{lname}]: no output: {msg}', flush=True)
                continue
            y = y.reshape(ref.shape)
            nan = np.isnan(y)
            d = np.where(nan, 1e9, np.abs(y - ref))            # (N, C, 1, T)
            per_line = d.max(axis=(1, 2, 3))
            bad = np.nonzero(per_line > 1e-3)[0]
            first_t = [(int(np.nonzero(d[i].max(axis=(0, 1)) > 1e-3)[0][0]), int((d[i].max(axis=(0, 1)) > 1e-3).sum())) for i in bad[:16]]
            if len(bad):
                i = int(bad[0])
                print('   line', i, 'wp :', np.array2string(y[i, :6, 0, :].T, precision=3, max_line_width=200))
                print('   line', i, 'ref:', np.array2string(ref[i, :6, 0, :].T, precision=3, max_line_width=200))
            print(f'{name} N={N} T={T} [{lname}]: nan={int(nan.sum())} maxdiff(finite)={float(np.abs(np.where(nan, 0, y - ref)).max()):.2e} '
                  f'bad lines {len(bad)}: {bad[:24].tolist()} first bad t {first_t} | {msg}', flush=True)
