"""Dev tool (GPU): mm_rpred on 1-channel bbox lines, device dewarp vs host dewarp: which records differ, and did the network see the same input?"""
import sys, warnings
from collections import defaultdict
sys.path.insert(0, '.')
import numpy as np, torch
from PIL import Image
from kraken_amd import rpred as R
from kraken_amd.containers import BBoxLine, Segmentation
from kraken_amd.models import TorchSeqRecognizer
from tests.helpers import build_model
from tests.specs import BENCH_A, bench_codec
from tests.helpers import wavy_line as _wavy_line

m = build_model(BENCH_A, codec=bench_codec(), seed=0)
m.seg_type, m.model_type = 'bbox', ['recognition']
net = TorchSeqRecognizer(m, device='cuda')
rng = np.random.RandomState(9)
rows, boxes, y = [], [], 0
for i in range(60):
    h, w = int(rng.randint(30, 90)), int(rng.randint(200, 1000))
    line = _wavy_line(rng, h, w) if i != 17 else np.full((h, w), 255, np.uint8)
    rows.append(np.pad(line, ((0, 0), (0, 1000 - w)), constant_values=255))
    boxes.append((0, y, w, y + h))
    y += h
page = Image.fromarray(np.vstack(rows), 'L')
seg = Segmentation(type='bbox', imagename='p', text_direction='horizontal-lr', script_detection=False,
                   lines=[BBoxLine(id=f'l{i}', bbox=list(b)) for i, b in enumerate(boxes)])
cap = {'crops': {}, 'hostfall': set(), 'tensors': {}}
sd, su = R.LinePipeline.submit_dewarp, R.LinePipeline.submit
def submit_dewarp(self, items, pad):
    for k, a in items: cap['crops'][k] = a.copy()
    widths, host = sd(self, items, pad)
    cap['hostfall'] |= set(host)
    cap.setdefault('widths', {}).update(widths)
    return widths, host
def submit(self, items, *a, **k):
    for key, t in items: cap['tensors'][key] = t.clone()
    return su(self, items, *a, **k)
R.LinePipeline.submit_dewarp, R.LinePipeline.submit = submit_dewarp, submit
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    dev = list(R.mm_rpred(defaultdict(lambda: net), page, seg, bidi_reordering=False))
    devcap = cap
    cap = {'crops': {}, 'hostfall': set(), 'tensors': {}}
    R.DEVICE_DEWARP = False
    host = list(R.mm_rpred(defaultdict(lambda: net), page, seg, bidi_reordering=False))
print('device run: lines dewarped on device', len(devcap['crops']), 'host fallbacks', sorted(devcap['hostfall']), 'host tensors', len(devcap['tensors']))
print('host run: tensors', len(cap['tensors']))
for i, (a, b) in enumerate(zip(dev, host)):
    ca, cb = np.array(a.confidences), np.array(b.confidences)
    same = a.prediction == b.prediction and len(ca) == len(cb) and (len(ca) == 0 or np.abs(ca - cb).max() < 1e-6)
    if not same:
        ht = cap['tensors'].get(i)
        w_dev = devcap.get('widths', {}).get(i)
        print(f'line {i}: box {boxes[i]} pred equal {a.prediction == b.prediction} max dconf {np.abs(ca - cb).max() if len(ca) == len(cb) and len(ca) else None} '
              f'host tensor {None if ht is None else tuple(ht.shape)} device width {w_dev} fallback {i in devcap["hostfall"]}')

# the same crops through the engine's two dewarp calls, in the batches mm_rpred formed: where do the network inputs differ?
from kraken_amd.engine import RecognitionEngine
from kraken_amd.transforms import ImageInputTransforms
ts = ImageInputTransforms(1, 48, 0, 1, (16, 0), valid_norm=True)
m2 = build_model(BENCH_A, codec=bench_codec(), seed=0).to('cuda')
eng = RecognitionEngine(m2, device=0, max_batch=64, max_width=2048, slots=1)
keys = sorted(devcap['crops'])
for lo in range(0, len(keys), 32):
    part = keys[lo:lo + 32]
    crops = [devcap['crops'][k] for k in part]
    r, ok, ink = eng.measure_dewarp(crops)
    use = ok & ink
    ticket = eng.submit_dewarped(r, use, 16)
    slot = eng.slots[ticket]
    slot.stream.synchronize()
    x = slot.keep.cpu()
    eng.collect(ticket)
    for k, key in enumerate(part):
        if not use[k]:
            continue
        hst = ts(Image.fromarray(crops[k], 'L'))
        wk = hst.shape[2]
        d = (x[k, :, :, :wk] - hst).abs()
        if float(d.max()) > 0:
            bad = torch.nonzero(d[0] > 0)
            print(f'line {key}: crop {crops[k].shape} r {int(r[k])}: {len(bad)} pixels differ, max {float(d.max()):.4f}, columns {sorted(set(bad[:, 1].tolist()))[:12]}, rows {sorted(set(bad[:, 0].tolist()))[:12]}')
eng.close()
