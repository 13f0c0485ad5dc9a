import torch
n = 245_760_000 // 4
x = torch.empty(n, device='cuda'); y = torch.empty(n, device='cuda')
def t(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True)
    a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / k
ms = t(lambda: x.zero_()); print('fill  %.3f ms  %.2f TB/s write' % (ms, n*4/ms/1e9))
ms = t(lambda: y.copy_(x)); print('copy  %.3f ms  %.2f TB/s r+w' % (ms, 2*n*4/ms/1e9))
ms = t(lambda: x.sum()); print('sum   %.3f ms  %.2f TB/s read' % (ms, n*4/ms/1e9))
