"""Pinned host -> device copy rate of this box, for the sizes the API path uploads per chunk of lines (a band of the page: ~43 MB of an
'L' page, ~172 MB of a colour page as R,G,B,X).  python tools/h2d_probe.py   (GPU)"""
import torch

for mb in (43, 172, 459):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(n, dtype=torch.uint8, device='cuda')
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            d.copy_(h, non_blocking=True)
        s.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record(s)
        for _ in range(5):
            d.copy_(h, non_blocking=True)
        b.record(s)
        s.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f'{mb:4d} MB pinned -> device: {ms:7.2f} ms = {n / ms / 1e6:6.1f} GB/s')
