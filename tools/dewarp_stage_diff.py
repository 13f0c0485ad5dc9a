"""Dev tool (GPU): krk_dewarp_measure against oracle/np_oracle.py stage by stage on the lines of tools/dewarp_api_diff.py."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from kraken_amd.engine import RecognitionEngine
from oracle import np_oracle as O
from tests.helpers import build_model
from tests.specs import BENCH_A, bench_codec
from tests.helpers import wavy_line as _wavy_line

rng = np.random.RandomState(9)
crops = []
for i in range(60):
    h, w = int(rng.randint(30, 90)), int(rng.randint(200, 1000))
    crops.append(_wavy_line(rng, h, w) if i != 17 else np.full((h, w), 255, np.uint8))
m = build_model(BENCH_A, codec=bench_codec(), seed=0).to('cuda')
eng = RecognitionEngine(m, device=0, max_batch=64, max_width=2048, slots=1)
for lo, hi in ((0, 32), (32, 60), (3, 4), (50, 51)):
    part = crops[lo:hi]
    n = len(part)
    r, ok, ink = eng.measure_dewarp(part)
    slot = [s for s in eng.slots if getattr(s, 'dw', None)][0] if not hasattr(eng, '_last') else None
    st = slot.dw
    torch.cuda.synchronize()
    maxw = st['maxw']
    work = st['work'].cpu().numpy()
    ridge = work[2 * n:2 * n + n * maxw].reshape(n, maxw)
    centre = work[2 * n + n * maxw:2 * n + 2 * n * maxw].reshape(n, maxw)
    desc = st['host_desc']
    scr = st['scratch'].cpu().numpy()
    for k, a in enumerate(part):
        if a.max() == a.min():
            continue
        h, w = a.shape
        line = a.astype(np.float64)
        top = line.max()
        inkk = (top - line) * 1.0 / (top - line).max()
        w0, r0 = O._gauss_weights(h * 0.5)
        w1, r1 = O._gauss_weights(h * 1.0)
        g0 = O._correlate_sym(inkk, w0, r0, 0, 'constant')
        blur = O._correlate_sym(g0, w1, r1, 1, 'constant')
        u0 = O._uniform_1d(blur, int(h * 0.5), 0)
        uni = O._uniform_1d(u0, int(w), 1)
        tot = blur + 0.001 * uni
        rg = np.argmax(tot, axis=0)
        ce = O.line_centers_np(inkk)
        soff = int(desc[k, 3])
        dblur = scr[soff + h * w:soff + 2 * h * w].reshape(h, w)
        duni = scr[soff + 2 * h * w:soff + 3 * h * w].reshape(h, w)
        du0 = scr[soff:soff + h * w].reshape(h, w)
        msg = []
        if not np.array_equal(dblur, blur): msg.append(f'blur differs in {int((dblur != blur).sum())} cells, max {np.abs(dblur - blur).max():.3e}')
        if not np.array_equal(du0, u0): msg.append(f'unif0 differs in {int((du0 != u0).sum())} cells, max {np.abs(du0 - u0).max():.3e}')
        if not np.array_equal(duni, uni): msg.append(f'unif1 differs in {int((duni != uni).sum())} cells, max {np.abs(duni - uni).max():.3e}')
        if not np.array_equal(ridge[k, :w], rg): msg.append(f'ridge differs at columns {np.nonzero(ridge[k, :w] != rg)[0][:8].tolist()}')
        if not np.array_equal(centre[k, :w], ce): msg.append(f'centre differs at columns {np.nonzero(centre[k, :w] != ce)[0][:8].tolist()}')
        if msg:
            print(f'batch [{lo},{hi}) line {lo + k} {a.shape}:', '; '.join(msg), flush=True)
print('done')
eng.close()
