"""Throughput of BENCH-B (the GroupNorm recogniser of the fixtures) in both plans, same engine/protocol as bench.py."""
import sys, time, torch
sys.path.insert(0, '.')
import kraken_amd
from kraken_amd.engine import RecognitionEngine
from kraken_amd.specs import BENCH_B, bench_codec
torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=BENCH_B, codec=bench_codec()).to('cuda')
x = torch.rand(256, 1, 48, 1200, generator=torch.Generator().manual_seed(1)).cuda()
for prec in ('f32', 'bf16x3'):
    m.nn.set_precision(prec)
    eng = RecognitionEngine(m, device=0, max_batch=256, max_width=1200, slots=4)
    def run(k):
        for _ in range(k):
            if eng.free_slots() == 0: eng.collect()
            eng.submit(x)
        while eng.free_slots() < 4: eng.collect()
    run(6); torch.cuda.synchronize(); t = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t
    eng.set_profiling(True); run(4)
    names = {}
    for n_, ms, _ in eng.layer_times()[0]: names[n_] = names.get(n_, 0) + ms
    print(prec, '%.0f lines/s, %.2f ms/step' % (256 * 40 / dt, 1e3 * dt / 40), {k: round(v, 2) for k, v in names.items()}, flush=True)
    eng.close()
    # the same steps with ONE batch in flight (each kernel has the chip to itself), in schedule order
    solo = RecognitionEngine(m, device=0, max_batch=256, max_width=1200, slots=1)
    solo.set_profiling(True)
    for _ in range(3):
        solo.submit(x)
        solo.collect()
    print(prec, 'alone:', ' '.join('%s=%.3f' % (n_, ms) for n_, ms, _ in solo.layer_times()[0]), flush=True)
    solo.close()
