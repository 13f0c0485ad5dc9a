"""Host AddressSanitizer driver (no GPU): runs the C ABI's C++ side over many networks, batch shapes and call orders against
kraken_amd/libkraken_amd_asanhost.so (python -m kraken_amd.build --asan-host: capi.hip instrumented, the HIP runtime replaced by
tools/asan/fake_hip.cpp -- device memory is malloc'ed host memory, launches are no-ops, NOTHING is computed).  What ASan watches:
the plan compiler, every weight packer's writes and uploads, length / shape arithmetic, workspace sizing and reuse, clone / destroy
orders, the decode wrappers' argument handling.  Started by tests/test_asan_host.py with clang's ASan runtime preloaded:
    LD_PRELOAD=$(clang -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_host_driver.py
Prints 'ASAN-HOST OK <plans> plans, <calls> calls, <launches> launches' and exits 0, or dies with ASan's report."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['KRAKEN_AMD_LIB'] = os.path.join(ROOT, 'kraken_amd', 'libkraken_amd_asanhost.so')

import numpy as np  # noqa: E402
import torch  # noqa: E402

import kraken_amd  # noqa: E402
from kraken_amd import _lib  # noqa: E402
from kraken_amd.specs import BENCH_A, BENCH_A_RGB, BENCH_B, DEFAULT_H120  # noqa: E402
from kraken_amd.vgsl import _Plan  # noqa: E402

lib = _lib.load()
assert lib.krk_device_count() == 1
fake = C.CDLL(os.environ['KRAKEN_AMD_LIB'])
fake.fake_hip_launches.restype = C.c_long
fake.fake_hip_live_allocations.restype = C.c_long

specs = [BENCH_A, BENCH_B, BENCH_A_RGB, DEFAULT_H120,
         '[1,48,0,1 Cr3,3,8 Mp2,2 S1(1x0)1,3 Lbx40 Lfx24 Lrx16 O1c12]',
         '[1,48,0,1 Cr3,3,16 Gn4 Mp2,2 Cr3,3,16 S1(1x0)1,3 Gbx32 O1c9]',
         '[1,16,0,1 Cr3,13,32 Mp2,2 Cr3,13,32 Mp2,2 Cr3,9,64 S1(1x0)1,3 Lbx72 O1a20]',
         '[1,32,0,3 (Cr3,3,8 Cr5,5,8) Mp2,2 S1(1x0)1,3 Lbx16 O1c7]',
         '[1,1,0,16 Lbx1024 O1c8]', '[1,1,0,16 Lfx1280 O1c8]', '[1,1,0,16 Lbxo832 O1c8]', '[1,1,0,16 Lbx264 O1c8]',
         '[1,24,0,1 Cr3,3,8 Lbx8 Lby8 Cr1,1,4 S1(1x0)1,3 O1c5]',
         '[1,48,0,1 Cr3,3,8 CTr2,2,6,2,2 Mp2,2 Mp2,2 S1(1x0)1,3 Lbx12 O1c6]',
         '[1,48,0,1 Cr3,3,12 A1,4 Mp2,2 S1(1x0)1,3 Lbx12 O1c6]',
         # round 6: zero filters appended on request (the plan is compiled again per request), odd hidden sizes (padded directions, any K),
         # 257-512 units, a first layer of 64 filters / 7 kernel rows / colour with 5, a stack the split kernels leave with the switch off
         '[1,48,0,1 Cr3,3,24 Mp2,2 Cr3,3,48 Mp2,2 Cr3,3,40 S1(1x0)1,3 Lbx64 O1c12]',
         '[1,48,0,1 Cr3,13,20 Mp2,2 Cr3,13,20 Mp2,2 Cr3,9,40 Mp2,2 Cr3,9,40 S1(1x0)1,3 Lbx100 Lbx150 Lfx75 O1c12]',
         '[1,48,0,1 Cr3,3,36 Cr3,3,20 Mp2,2 Cr5,5,44 Mp2,2 Cr3,3,12 S1(1x0)1,3 Lbx6 Lfx5 Lrx3 O1c5]',
         '[1,48,0,1 Cr3,3,64 Mp2,2 Cr3,3,32 S1(1x0)1,3 Lbx300 Lbx8 O1c5]',
         '[1,48,0,1 Cr7,7,32 Mp2,2 Cr3,3,64 Mp2,2 S1(1x0)1,3 Lbx200 O1c80]',
         '[1,48,0,3 Cr5,5,32 Mp2,2 Cr3,3,64 S1(1x0)1,3 Lfx75 O1c5]']
try:        # the 80 random specs whose state-dict names are pinned against the reference: a wide sample of nested groups
    names = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'spec_names.json')))
    specs += [c['spec'] for c in (names.values() if isinstance(names, dict) else names)][:40]
except Exception:
    pass

rng = np.random.default_rng(5)
plans = calls = skipped = 0
for spec in specs:
    torch.manual_seed(0)
    try:
        m = kraken_amd.TorchVGSLModel(vgsl=spec)
    except (NotImplementedError, ValueError):
        skipped += 1
        continue
    _, c, h, _ = m.nn._input
    if h <= 0:
        h = 32
    for prec in (_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16):
        try:
            plan = _Plan(m.nn._specs if h == m.nn._input[2] else m.nn._specs_for_height(h), m.nn, c, h, 0, prec)
        except _lib.KrakenAmdError:
            continue                                   # a network this arithmetic does not cover: refused with a message, not a crash
        plans += 1
        twin = plan.clone()
        for N, W in ((1, 37), (3, 200), (17, 403), (2, 9)):
            x = np.ascontiguousarray(rng.random((N, c, h, W), dtype=np.float32))
            lens = np.sort(rng.integers(1, W + 1, N)).astype(np.int32)[::-1].copy()
            lens[0] = W
            for p, use_lens in ((plan, True), (twin, False)):
                try:
                    n_out, co, ho, wo = p.out_dims(N, W)
                except _lib.KrakenAmdError:
                    continue                           # a Reshape that does not divide this (N, W)
                out = np.empty(max(1, n_out * co * ho * wo) + 64, dtype=np.float32)
                rc = lib.krk_forward(p.handle, x.ctypes.data, lens.ctypes.data if use_lens else None, N, W, None, out.ctypes.data)
                calls += 1
                if rc == 0 and use_lens:
                    try:
                        p.olens(lens, W)
                    except _lib.KrakenAmdError:
                        pass
                if rc == 0 and ho == 1 and n_out == N:
                    T = wo
                    nt = N * T
                    buf = np.empty(4 * nt + N, dtype=np.int32)
                    base = buf.ctypes.data
                    dec = _lib.KrkDecodeOut(base, base + 4 * nt, base + 8 * nt, base + 12 * nt, base + 16 * nt, T)
                    olens = np.empty(N, dtype=np.int32)
                    probs = np.empty(nt * co, dtype=np.float32)
                    lib.krk_recognize(p.handle, x.ctypes.data, lens.ctypes.data if use_lens else None, N, W, C.c_float(1.0), None, None,
                                      probs.ctypes.data, olens.ctypes.data, C.byref(dec))
                    calls += 1
            lib.krk_plan_status(plan.handle)
        # destroy in both orders: the packed weights belong to whichever plan goes last
        if plans % 2:
            plan.close(); twin.close()
        else:
            twin.close(); plan.close()
live = fake.fake_hip_live_allocations()
assert live == 0, f'{live} device allocations were never freed'
print(f'ASAN-HOST OK {plans} plans ({skipped} specs refused at construction), {calls} calls, {fake.fake_hip_launches()} launches')
