"""Dev tool (GPU): what the FIRST host -> device copies of a fresh process cost (cold-start work).
Times 15 x 8 MB pinned -> device copies (async + one sync) three times over, with / without touching the destination first."""
import sys
import time

import torch

torch.cuda.init()
t0 = time.perf_counter()
torch.zeros(1, device='cuda')
torch.cuda.synchronize()
print('context + first kernel %.1f ms' % (1e3 * (time.perf_counter() - t0)))
piece = 8 << 20
variant = sys.argv[1] if len(sys.argv) > 1 else 'plain'
ring = [torch.empty(piece, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
for r in ring:
    r.fill_(1)
st = torch.cuda.Stream()
for rep in range(3):
    t0 = time.perf_counter()
    dev = torch.empty(15 * piece, dtype=torch.uint8, device='cuda')
    t1 = time.perf_counter()
    if variant == 'touch':
        dev.zero_()
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    per = []
    with torch.cuda.stream(st):
        for k in range(15):
            ta = time.perf_counter()
            dev[k * piece:(k + 1) * piece].copy_(ring[k % 4], non_blocking=True)
            per.append(1e3 * (time.perf_counter() - ta))
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f'{variant} rep {rep}: alloc {1e3 * (t1 - t0):.2f} touch {1e3 * (t2 - t1):.2f} issue {1e3 * (t3 - t2):.2f} '
          f'(per copy max {max(per):.2f} min {min(per):.2f}) drain {1e3 * (t4 - t3):.2f} ms -> {15 * piece / (t4 - t2) / 1e9:.1f} GB/s')
    if variant == 'keep':
        pass
    else:
        del dev
