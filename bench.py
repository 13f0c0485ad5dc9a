#!/usr/bin/env python
"""
Benchmark of the kraken line-recognition hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (VGSL conv stack + 3 BiLSTM + linear + softmax + CTC
best-path decode, label tuples copied back to the host, host codec -> strings) over ONE batch of 256 synthetic
1x48x1200 line images per GPU (BASELINE.json configs[1]; BENCH-A spec of SURVEY.md section 8d, random-init
weights `torch.manual_seed(0)`).  Inputs are resident in HBM before the timed region.  Ranks
shard lines (weak scaling: 256 lines per rank per step) and the decoded label sequences are
gathered on all ranks over RCCL after the last step of the timed region.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : dominant kernel (the f32-MFMA implicit-GEMM convolution) -- algorithmic FLOPs of one
                 launch / its mean duration from HIP events on its own stream, vs the f32 MFMA peak
  cpu_baseline : the reference's PyTorch-CPU path (oracle/torch_port.py, kind "port") timed on this
                 box's host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # one hardware queue per engine slot (read when HIP starts)
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16, dense (not the 2:1-sparse figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--batch', type=int, default=256, help='lines per GPU per step')
    ap.add_argument('--width', type=int, default=1200)
    ap.add_argument('--slots', type=int, default=4, help='batches in flight per GPU (streams)')
    ap.add_argument('--precision', default='bf16x3', choices=['f32', 'bf16x3'],
                    help='f32: exact f32 MFMA; bf16x3: split-bf16 operands on the bf16 MFMA, f32 accumulate (fp32-class)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-input', action='store_true',
                    help='hand every batch over as a pinned HOST tensor (PCIe-inclusive rate; DESIGN.md quotes it, `value` never does)')
    ap.add_argument('--force-dist', action='store_true', help='initialise RCCL and run the gather even with one rank (smoke test)')
    ap.add_argument('--cpu-lines', type=int, default=32, help='lines in the CPU baseline sample')
    return ap.parse_args()


def cpu_baseline(model, width, n_lines, reps=2):
    """kraken's CPU path (torch CPU operators + Python greedy decode + codec), bounded sample."""
    from oracle.torch_port import CpuRecognizer
    ref = CpuRecognizer(model.layer_specs, {k: v.cpu() for k, v in model.state_dict().items()})
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(n_lines, 1, 48, width, generator=g)
    lens = [width] * n_lines
    ref.predict_labels(x[:2], lens[:2])   # warm-up
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        tuples = ref.predict_labels(x, lens)
        _ = [''.join(c for c, *_ in model.codec.decode(t)) for t in tuples]
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {'value': round(n_lines / best, 2), 'unit': 'lines/s', 'cores': torch.get_num_threads(),
            'host_cpus': os.cpu_count(), 'kind': 'port',
            'sample': f'{n_lines} lines 1x48x{width}, fp32, best of {reps}: torch-CPU forward + softmax + '
                      f'groupby greedy decode + codec (oracle/torch_port.py = kraken lib/models.py:138-149)'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    import kraken_amd
    from kraken_amd import _lib, dist as kdist
    from kraken_amd.engine import RecognitionEngine
    from tests.specs import BENCH_A, bench_codec

    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device(f'cuda:{local_rank}')
    use_dist = world > 1 or args.force_dist
    if use_dist:
        kdist.init(backend='nccl')

    torch.manual_seed(0)
    model = kraken_amd.TorchVGSLModel(vgsl=BENCH_A, codec=bench_codec())
    model.to(dev)
    model.nn.set_precision(args.precision)
    N, W = args.batch, args.width
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(N, 1, 48, W, generator=g).to(dev)    # resident in HBM before timing
    engine = RecognitionEngine(model, device=local_rank, max_batch=N, max_width=W, slots=args.slots)
    if args.host_input:
        x = x.cpu().pin_memory()

    codec = model.codec
    n_chars = [0]

    def finish():
        batch, olens = engine.collect()
        strings = codec.decode_strings(batch)          # host codec: label tuples -> text, inside the timed region
        n_chars[0] += sum(map(len, strings))
        return batch, olens

    def run(steps):
        last = None
        for _ in range(steps):
            if engine.free_slots() == 0:
                last = finish()
            engine.submit(x)
        while engine.free_slots() < len(engine.slots):
            last = finish()
        return last

    run(args.warmup)
    engine.set_profiling(True)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    batch, olens = run(args.steps)
    gathered = kdist.gather_decoded(batch, olens, force=args.force_dist) if use_dist else [batch]
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-launch timing from the HIP events recorded inside the timed region (last batch of each
    # slot, on the stream the kernels were launched on)
    per_launch = {}
    for slot_times in engine.layer_times():
        for i, (name, ms, flops) in enumerate(slot_times):
            e = per_launch.setdefault(i, {'name': name, 'ms': [], 'flops': flops})
            e['ms'].append(ms)
    launches = [{'i': i, 'name': v['name'], 'ms': float(np.mean(v['ms'])), 'gflop': v['flops'] / 1e9}
                for i, v in sorted(per_launch.items())]
    # kernels behind the launch groups (rocprofv3 kernel names)
    kernel_of = {'conv': 'conv_f32_kernel', 'lstm_xproj': 'conv_f32_kernel<1,1,0,4>', 'linear': 'conv_f32_kernel<1,1,0,4>',
                 'lstm_rec': 'lstm_f32_kernel', 'conv_x3': 'conv_x3_kernel', 'conv1_x3': 'conv1_x3_kernel',
                 'conv_taps_x3': 'conv_taps_kernel', 'lstm_xproj_x3': 'gemm_x3_kernel', 'linear_x3': 'gemm_x3_kernel',
                 'lstm_rec_x3': 'lstm_x3_kernel'}
    peak_of = lambda name: BF16_MFMA_PEAK_TFLOPS if name.endswith('_x3') else F32_MFMA_PEAK_TFLOPS   # noqa: E731
    groups = {}
    for l in launches:
        g = groups.setdefault(l['name'], {'ms': 0.0, 'gflop': 0.0, 'n': 0})
        g['ms'] += l['ms']
        g['gflop'] += l['gflop']
        g['n'] += 1
    # dominant kernel = the launch group with the largest share of the step
    dom_name = max((k for k in groups if groups[k]['gflop'] > 0), key=lambda k: groups[k]['ms'])
    dom = groups[dom_name]
    ach = dom['gflop'] / dom['ms']   # GFLOP/ms == TFLOP/s; algorithmic FLOPs of the launches / their duration
    roofline = {'bound': 'mfma', 'kernel': kernel_of.get(dom_name, dom_name), 'launch_group': dom_name,
                'launches_per_step': dom['n'], 'avg_launch_ms': round(dom['ms'] / dom['n'], 4),
                'gflop_per_launch': round(dom['gflop'] / dom['n'], 3),
                'achieved': round(ach, 2), 'peak': peak_of(dom_name), 'unit': 'TFLOP/s',
                'frac': round(ach / peak_of(dom_name), 4), 'traffic': None,
                'note': ('algorithmic FLOPs; the split-operand kernels issue 3 bf16 MFMAs per algorithmic product'
                         if dom_name.endswith('_x3') else 'exact f32 MFMA')}
    layers = launches
    # HBM traffic of the dominant kernel from the committed PMC summary of this same command (separate
    # rocprofv3 --pmc passes, see tools/summarize_pmc.py); null when no summary is available
    try:
        tag = 'r01_bf16x3' if args.precision == 'bf16x3' else 'r01'
        pmc = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_summary.json')))['kernels']
        kname = roofline['kernel'].split('<')[0]
        hit = [v for k, v in pmc.items() if k.startswith(kname) and 'hbm_write_MB_per_launch' in v]
        if hit:
            v = max(hit, key=lambda e: e.get('total_ms', 0))
            rd = v.get('hbm_read_MB_x2', v.get('hbm_read_MB_per_launch', 0.0))
            roofline['traffic'] = {'read_MB': rd, 'write_MB': v['hbm_write_MB_per_launch'], 'per': 'launch',
                                   'source': f'profiles/{tag}_pmc_summary.json (FETCH_SIZE/WRITE_SIZE passes)'}
    except Exception:
        pass

    lines = N * args.steps * world
    value = lines / dt
    out = {
        'metric': 'text lines/sec (whole node) at 48x1200px, VGSL CNN+BiLSTM+CTC',
        'value': round(value, 1), 'unit': 'lines/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32' if args.precision == 'f32' else 'bf16x3 (every value carried as bf16 hi+lo; 3 bf16 MFMAs per product, f32 accumulate; |d logit| vs fp32 ~1.4e-5)',
        'data': 'synthetic' + (' (pinned host input per step: PCIe-inclusive)' if args.host_input else ''),
        'config': {'workload': f'BENCH-A VGSL recogniser (3.17M params, random init seed 0), {N} lines 1x48x{W} per GPU '
                               f'per step, greedy CTC decode, label tuples to host', 'lines_per_gpu_step': N, 'width': W,
                   'slots': args.slots, 'precision': args.precision, 'parallelism': f'dp{world}', 'whole_path_tflops': round(value * 2.778e-3 *
                                                                                                 (W / 1200.0), 2)},
        'roofline': roofline,
        'launches': [{'name': l['name'], 'ms': round(l['ms'], 3), 'tflops': round(l['gflop'] / l['ms'], 1) if l['ms'] > 0 else 0}
                     for l in layers],
        'groups': {k: {'ms': round(v['ms'], 3), 'tflops': round(v['gflop'] / v['ms'], 1) if v['ms'] > 0 else 0,
                       'frac_of_peak': round(v['gflop'] / v['ms'] / peak_of(k), 4) if v['ms'] > 0 else 0}
                   for k, v in groups.items()},
        'gathered_lines': int(sum(len(b.counts) for b in gathered)),
        'decoded_chars': int(n_chars[0]),
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(model, W, args.cpu_lines)
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
