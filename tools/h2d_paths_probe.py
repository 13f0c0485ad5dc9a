"""Dev tool (GPU): three ways to get a 460 MB R,G,B,X page from pageable host memory (Pillow's blocks) to the device --
(a) memmove on 8 threads into a pinned buffer + one asynchronous copy (what the API path does), (b) one pageable copy_ of the source
(the runtime stages it), (c) hipHostRegister the source, copy asynchronously, unregister.  python tools/h2d_paths_probe.py [MB]"""
import ctypes
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 460
n = mb << 20
torch.cuda.init()
rt = torch.cuda.cudart()
dev = torch.empty(n, dtype=torch.uint8, device='cuda')
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
pool = ThreadPoolExecutor(8)
torch.cuda.synchronize()
for rep in range(3):
    src = np.random.default_rng(rep).integers(0, 255, n, dtype=np.uint8)      # fresh pageable memory, like a freshly decoded image
    st = torch.from_numpy(src)
    # (a)
    t0 = time.perf_counter()
    step = n // 64
    base_d, base_s = pin.numpy().ctypes.data, src.ctypes.data
    list(pool.map(lambda k: ctypes.memmove(base_d + k * step, base_s + k * step, step), range(64)))
    t1 = time.perf_counter()
    dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # (b)
    dev.copy_(st)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    # (c)
    rc = rt.cudaHostRegister(src.ctypes.data, n, 0)
    t4 = time.perf_counter()
    dev.copy_(st, non_blocking=True)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    rt.cudaHostUnregister(src.ctypes.data)
    t6 = time.perf_counter()
    print(f'rep {rep}: (a) memmove {1e3 * (t1 - t0):.1f} + dma {1e3 * (t2 - t1):.1f} ms | (b) pageable copy {1e3 * (t3 - t2):.1f} ms | '
          f'(c) register {1e3 * (t4 - t3):.1f} (rc {int(rc)}) + copy {1e3 * (t5 - t4):.1f} + unregister {1e3 * (t6 - t5):.1f} ms', flush=True)
