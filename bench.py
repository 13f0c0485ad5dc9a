#!/usr/bin/env python
"""
Benchmark of the kraken line-recognition hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Default mode (the driver's contract).  One "step" = one pass of the hot path (VGSL conv stack + 3 BiLSTM + linear +
softmax + CTC best-path decode, label tuples copied back to the host, host codec -> strings) over ONE batch of 256
synthetic 1x48x1200 line images per GPU (BASELINE.json configs[1]; BENCH-A spec of SURVEY.md section 8d, random-init
weights `torch.manual_seed(0)`).  Inputs are resident in HBM before the timed region (four distinct batches, rotated:
236 MB, so a step never re-reads an input the previous step left in the last-level cache).  Ranks shard lines (weak
scaling: 256 lines per rank per step); the decoded label sequences of EVERY step are gathered on all ranks over RCCL in
one exchange at the end of the timed region (BASELINE config 3: "gather of decoded strings").

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : the launch group with the largest share of the step -- algorithmic FLOPs of its launches / their mean
                 duration from HIP events on the stream they ran on, vs the dense MFMA peak of its operand type
  cpu_baseline : the reference's PyTorch-CPU path (oracle/torch_port.py, kind "port") timed on this box's host cores on
                 a bounded sample (rank 0, N=1 only): best of a thread sweep, plus the legacy one-line-per-call shape

Other modes (secondary measurements, same JSON shape, `config.workload` says which):
  --mode api      lines/s through the REFERENCE API (kraken_amd.rpred.rpred generator: crop -> transform -> network ->
                  records) on one synthetic page of --api-lines bbox lines, next to the resident-input engine number
  --mode config4  BASELINE config 4: 1024 lines, widths U{400..2400}, width-bucketed batches through the same pipeline
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
METRIC = 'text lines/sec (whole node) at 48x1200px, VGSL CNN+BiLSTM+CTC'
DTYPE_X3 = 'bf16x3 (every value carried as bf16 hi+lo; 3 bf16 MFMAs per product, f32 accumulate; |d logit| vs fp32 ~1.4e-5)'
DTYPE_BF16 = 'bf16 (OPT-IN plan: plain bf16 operands, f32 accumulate; greedy strings identical on the fixtures, |d logit| ~1e-2 -- outside the 1e-3 parity gate)'
DTYPES = {'f32': 'f32', 'bf16x3': DTYPE_X3, 'bf16': DTYPE_BF16}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--preheat-ms', type=float, default=400.0,
                    help='untimed run of the same step BEFORE the W warm-up steps (default 400 ms; 0 = none): the process has just '
                         'built its plans on an idle GPU, whose clock governor needs more than W = 5 steps (12 ms) to leave its idle state; '
                         'a service is never in that state.  Reported in the line as `preheat`')
    ap.add_argument('--mode', default='engine', choices=['engine', 'api', 'config4'])
    ap.add_argument('--batch', type=int, default=256, help='lines per GPU per step')
    ap.add_argument('--width', type=int, default=1200)
    ap.add_argument('--slots', type=int, default=None, help='batches in flight per GPU (streams); default 3 (config4: 4, r5 sweep: 128-line buckets x 4 slots 83.8 k lines/s, 256 x 3 79.9 k); 3 since the recurrent cluster kernel halved the per-batch latency (r2: 93.7 k vs 87.4 k lines/s at 4 over 20 steps)')
    ap.add_argument('--precision', default='bf16x3', choices=['f32', 'bf16x3', 'bf16'],
                    help='f32: exact f32 MFMA; bf16x3 (headline): split-bf16 operands on the bf16 MFMA, f32 accumulate (fp32-class); '
                         'bf16: OPT-IN plain bf16 operands (strings-identical gate, logits ~1e-2: outside the parity gate, never the headline)')
    ap.add_argument('--data', default='noise', choices=['noise', 'strokes', 'dense'],
                    help="synthetic input set (SURVEY.md 8d): 'noise' = U(0,1) pixels (the headline); 'strokes' = zeros with 5 %% of the "
                         "columns set to U(0.5,1) (run lengths of real lines); 'dense' = lines of glyph cells through the same architecture with "
                         "rescaled LSTM input projections and recalibrated output biases, so that a line decodes to ~70 characters (the "
                         "host codec / record side at real text density)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-self-profile', action='store_true',
                    help='skip the rocprofv3 passes bench.py runs on itself after the timed region (kernel trace + three --pmc passes '
                         'of a 6-step run); roofline.traffic / rocprof_* then come from the newest committed summary under profiles/')
    ap.add_argument('--no-config4-check', action='store_true', help='skip the 1024 config-4 golden lines after the timed region')
    ap.add_argument('--host-input', action='store_true',
                    help='hand every batch over as a pinned HOST tensor (PCIe-inclusive rate; DESIGN.md quotes it, `value` never does)')
    ap.add_argument('--force-dist', action='store_true', help='initialise RCCL and run the gather even with one rank (smoke test)')
    ap.add_argument('--cpu-lines', type=int, default=32, help='lines in the CPU baseline sample')
    ap.add_argument('--stub-engine', action='store_true',
                    help='TEST ONLY (tests/test_dist_cpu.py): replace the device engine by a host stub and RCCL by gloo, to run the '
                         'launcher / sharding / gather plumbing on a box without GPUs; the JSON line says so and its value is meaningless')
    ap.add_argument('--share-device', action='store_true',
                    help='PLUMBING ONLY: all --gpus N ranks run on HIP device 0 (gloo collective: RCCL refuses two ranks per device). Two real '
                         'processes, two real engines, ShardedRecognizer.stream + gather, recognize_lines in input order -- on a one-GPU box. '
                         'The line is labelled as such and is NEVER a scaling number')
    ap.add_argument('--bucket-lines', type=int, default=128, help='--mode config4: most lines per device batch')
    ap.add_argument('--bucket-px', type=int, default=0, help='--mode config4: most padded pixels (lines x widest line) per device batch; 0 = no limit')
    ap.add_argument('--no-pcie', action='store_true', help='--mode config4: skip the host-image leg')
    ap.add_argument('--api-lines', type=int, default=2048, help='--mode api: bbox lines on the synthetic page')
    ap.add_argument('--api-workers', type=int, default=6, help='--mode api: host threads preparing lines (PIL conversions hold the GIL: more than ~6 threads only contend, 16 cost 40 %%)')
    args = ap.parse_args()
    if args.slots is None:
        args.slots = 4 if args.mode == 'config4' or args.precision == 'f32' else 3      # (f32 plan: its recurrences hold 32 CUs for 2 ms each: 4 in flight 31.6 k, 3: 28.7 k)
    return args


def synth_lines(n, width, seed, data='noise'):
    """One batch of the synthetic input sets of SURVEY.md section 8d: [n, 1, 48, width] fp32 in [0, 1]."""
    g = torch.Generator().manual_seed(seed)
    if data == 'dense':
        # "text": 60 glyph cells per 1200 px, each a 48 x 16 pattern out of a 96-glyph alphabet followed by 4 blank columns
        pitch, gap, K = 20, 4, 96
        glyphs = torch.rand(K, 48, pitch - gap, generator=g)
        idx = torch.randint(0, K, (n, width // pitch), generator=g)
        x = torch.zeros(n, 1, 48, width)
        for b in range(width // pitch):
            x[:, 0, :, b * pitch:b * pitch + pitch - gap] = glyphs[idx[:, b]]
        return x
    x = torch.rand(n, 1, 48, width, generator=g)          # 'noise': the tensor tests/golden/bench_lines.npz pins for seed 1234
    if data == 'strokes':
        on = torch.rand(n, 1, 1, width, generator=g) < 0.05
        x = torch.where(on, 0.5 + 0.5 * x, torch.zeros(()))
    return x


def densify(model, x, target=0.35, gain=10.0):
    """
    --data dense: a random-init recogniser decodes ~7 characters per 1200-px line whatever the input (random LSTMs are low-pass:
    the argmax hardly moves along the line); a real line of that width carries 60-80.  Same architecture, same kernels, same
    FLOPs, other numbers: the LSTM INPUT projections are scaled by `gain` (the gates then follow the glyph cells), every output
    class's bias becomes minus its mean logit over a calibration batch, and the blank's bias is raised until it wins `target` of
    the steps.  Deterministic for a seed; the CPU leg gets the same state dict.  ~70 characters per line on the glyph input.
    """
    with torch.no_grad():
        for name, p in model.nn.named_parameters():
            if 'weight_ih' in name:
                p.mul_(gain)
    model.nn.invalidate()
    _, _, logits, _ = model.nn.recognize(x, None, want_logits=True)          # [n, C, T]
    z = logits.float()
    mean = z.mean(dim=(0, 2))
    zc = z - mean[None, :, None]
    gap = (zc[:, 1:, :].max(dim=1).values - zc[:, 0, :]).flatten()
    lift = torch.quantile(gap, target)
    lin = [m for m in model.nn.modules() if hasattr(m, 'lin')][-1].lin
    with torch.no_grad():
        delta = -mean
        delta[0] += lift
        lin.bias.add_(delta.to(lin.bias.device, lin.bias.dtype))
    model.nn.invalidate()


def _kraken_recognizer(model):
    """kraken's own TorchSeqRecognizer over the SAME weights, when an installed kraken imports (never on the driver's GPU box:
    /root/reference does not travel there).  Returns None otherwise."""
    try:
        from kraken.lib import vgsl as kvgsl
        from kraken.lib.models import TorchSeqRecognizer as KRecognizer
        net = kvgsl.TorchVGSLModel(vgsl=model.spec, codec=model.codec.c2l)
        net.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
        net.eval()
        return KRecognizer(net, device='cpu')
    except Exception:
        return None


def cpu_baseline(model, width, n_lines, data='noise', batch=256):
    """
    kraken's CPU path (torch CPU operators + Python greedy decode + codec) on a bounded sample: the batched shape at the
    best intra-op thread count of a sweep, and the legacy rpred shape (one line per call).  kind 'reference' = an installed
    kraken's TorchSeqRecognizer.predict_string (kraken/lib/models.py:138-149) on the same weights; kind 'port' =
    oracle/torch_port.py, the restatement pinned to it.  Returns (record, strings of the sample).
    """
    kref = _kraken_recognizer(model)
    if kref is None:
        from oracle.torch_port import CpuRecognizer
        ref = CpuRecognizer(model.layer_specs, {k: v.cpu() for k, v in model.state_dict().items()})
    x = synth_lines(max(n_lines, batch), width, 1234, data)[:n_lines]       # == the first lines of the first timed batch of rank 0
    lens = [width] * n_lines

    def run(xs, ls):
        with torch.inference_mode():
            if kref is not None:
                return kref.predict_string(xs, torch.tensor(ls) if ls is not None else None)
            tuples = ref.predict_labels(xs, ls)
        return [''.join(c for c, *_ in model.codec.decode(t)) for t in tuples]

    ncpu = os.cpu_count() or 1
    saved = torch.get_num_threads()
    sweep, strings = {}, None
    for t in sorted({min(t, ncpu) for t in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(t)
        run(x[:2], lens[:2])                      # warm-up
        t0 = time.perf_counter()
        strings = run(x, lens)
        sweep[t] = n_lines / (time.perf_counter() - t0)
    best_t = max(sweep, key=sweep.get)
    torch.set_num_threads(best_t)
    t0 = time.perf_counter()
    k = min(8, n_lines)
    for i in range(k):                            # legacy shape: rpred calls the network with ONE line (kraken/rpred.py:226)
        run(x[i:i + 1], None)
    legacy = k / (time.perf_counter() - t0)
    torch.set_num_threads(saved)
    what = ('kraken.lib.models.TorchSeqRecognizer.predict_string of the installed kraken' if kref is not None else
            'oracle/torch_port.py = kraken lib/models.py:138-149')
    rec = {'value': round(sweep[best_t], 2), 'unit': 'lines/s', 'cores': best_t, 'host_cpus': ncpu,
           'kind': 'reference' if kref is not None else 'port',
           'thread_sweep': {str(t): round(v, 2) for t, v in sweep.items()},
           'legacy_one_line_per_call': round(legacy, 2),
           'sample': f'{n_lines} lines 1x48x{width} ({data}) in one batch, fp32, best of the thread sweep: torch-CPU forward + softmax + '
                     f'groupby greedy decode + codec ({what}); legacy = {k} lines, one per call'}
    if kref is None:
        rec['why_port'] = ('kraken itself is not importable on this box (pip install kraken, or put a checkout on PYTHONPATH, and this '
                           'leg runs kraken.lib.models.TorchSeqRecognizer instead: kind "reference"); the port runs the same torch-CPU '
                           'operators and is checked against kraken in the authoring container (tests/golden/make_golden.py, '
                           'tests/test_oracle_golden.py)')
    return rec, strings


# kernels behind the launch groups (rocprofv3 kernel names)
KERNEL_OF = {'conv': 'conv_f32_kernel', 'lstm_xproj': 'conv_f32_kernel<1,1,0,4>', 'linear': 'conv_f32_kernel<1,1,0,4>',
             'lstm_rec': 'lstm_f32_kernel', 'conv_x3': 'conv_x3', 'conv1_x3': 'conv1_x3_kernel',
             'conv_taps_x3': 'conv_taps_kernel', 'lstm_xproj_x3': 'gemm_x3_kernel', 'linear_x3': 'gemm_x3_kernel',   # (conv_x3 matches conv_x3p_kernel too)
             'lstm_rec_x3': 'lstm_ws_kernel'}


def roofline_of(engine, precision, fresh=None):
    """Per-launch timing from the HIP events recorded inside the timed region (last batch of each slot, on its own stream)."""
    per_launch = {}
    for slot_times in engine.layer_times():
        for i, (name, ms, flops) in enumerate(slot_times):
            e = per_launch.setdefault(i, {'name': name, 'ms': [], 'flops': flops})
            e['ms'].append(ms)
    launches = [{'i': i, 'name': v['name'], 'ms': float(np.mean(v['ms'])), 'gflop': v['flops'] / 1e9}
                for i, v in sorted(per_launch.items())]
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
    roofline = {'bound': 'mfma', 'kernel': KERNEL_OF.get(dom_name, dom_name), 'launch_group': dom_name,
                'launches_per_step': dom['n'], 'avg_launch_ms': round(dom['ms'] / dom['n'], 4),
                'gflop_per_launch': round(dom['gflop'] / dom['n'], 3),
                'achieved': round(ach, 2), 'peak': peak_of(dom_name), 'unit': 'TFLOP/s',
                'frac': round(ach / peak_of(dom_name), 4), 'traffic': None,
                'clock': 'HIP events on the stream the kernels ran on (rocprofv3: rocprof_avg_launch_ms / frac_rocprof, from the passes named in traffic.source)',
                'note': ('algorithmic FLOPs of the launch group / its HIP-event time with the other batches in flight; the '
                         'split-operand kernels issue 3 bf16 MFMAs per algorithmic product'
                         if dom_name.endswith('_x3') else 'exact f32 MFMA')}
    # rocprofv3 quantities of the SAME command from the committed PMC summary (separate --pmc passes at the benchmark's slot count,
    # tools/profile_round.sh + tools/summarize_pmc.py): the NEWEST summary of this precision under profiles/ is used and named in
    # `source`; everything below stays null / absent when there is none.  Per launch group: the kernel's rocprofv3 average duration,
    # MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles): of the chip, and of the CUs the kernel occupies),
    # HBM bytes per launch (FETCH_SIZE doubled where the guide's gfx950 half-count applies) and the HBM rate they imply.
    pmc_groups, step_traffic = {}, None
    try:
        import glob
        pat = 'r*_bf16x3_pmc_summary*.json' if precision != 'f32' else 'r*_f32_pmc_summary*.json'
        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', pat)) or
                       ([] if precision != 'f32' else glob.glob(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json'))))
        cands = [c for c in cands if 'solo' not in c] or cands
        if fresh and fresh.get('summary'):
            whole, tag = fresh['summary'], 'THIS RUN'
            src = ('this run: rocprofv3 passes bench.py launched on its own command after the timed region (' + fresh['command'] + '; '
                   '--kernel-trace --stats, then --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, one pass each)')
        else:
            summary = cands[-1]
            tag = os.path.basename(summary)
            whole = json.load(open(summary))
            src = f'profiles/{tag} (COMMITTED file, not this run' + (': ' + fresh['why'] if fresh and fresh.get('why') else '') + ')'
        pmc = whole['kernels']
        by_grid = (fresh or {}).get('by_grid') or {}
        # the clock the committed profile ran at (tools/clock_sample.py; the MFMA-busy fractions are of ACTUAL cycles)
        sclk = ((whole.get('clock') or {}).get('sclk_mhz') or {}).get('median')
        steps_profiled = max(1, max((v.get('calls', 0) for k, v in pmc.items() if k.startswith('rowmax_rows_kernel')), default=1))

        def entry(kname):
            hit = [v for k, v in pmc.items() if k.startswith(kname.split('<')[0]) and 'hbm_write_MB_per_launch' in v]
            return max(hit, key=lambda e: e.get('total_ms', 0)) if hit else None

        def cus_of(kname, v):          # CUs a launch occupies: the recurrent cluster kernels hold 4 CUs per 32 lines and direction
            return 64.0 if kname.startswith('lstm_ws') else 256.0

        for gname in groups:
            kname = KERNEL_OF.get(gname, gname)
            v = entry(kname)
            if not v:
                continue
            rd = v.get('hbm_read_MB_x2', v.get('hbm_read_MB_per_launch', 0.0))
            wr = v['hbm_write_MB_per_launch']
            e = {'kernel': kname, 'rocprof_avg_launch_ms': round(v['avg_us'] / 1e3, 4), 'hbm_read_MB': rd, 'hbm_write_MB': wr,
                 'hbm_GBps': round((rd + wr) / v['avg_us'] * 1e3, 1) if v.get('avg_us') else None}
            sharers = [g for g in groups if g != gname and KERNEL_OF.get(g, g) == kname]
            if sharers:
                # rocprofv3's stats average over ALL launches of a kernel; the kernel trace tells them apart by grid size, and a launch
                # group is the grid that occurs groups[g]['n'] times per step (three projections, one linear layer)
                mine = [g_ for g_ in by_grid.get(kname.split('<')[0], []) if g_['per_step'] == groups[gname]['n']]
                others = {groups[o]['n'] for o in sharers}
                if len(mine) == 1 and groups[gname]['n'] not in others:
                    e['rocprof_avg_launch_ms'] = round(mine[0]['avg_us'] / 1e3, 4)
                    e['grid'] = mine[0]['grid']
                    e['counters_averaged_with'] = sharers      # the --pmc passes still average the kernel's launches
                    e['hbm_GBps'] = None                       # (bytes of the average launch over this grid's duration would mean nothing)
                else:
                    e['averaged_with'] = sharers
            if v.get('mfma_util_chip') is not None:
                e['mfma_busy_chip'] = v['mfma_util_chip']
                e['mfma_busy_on_its_CUs'] = round(v['mfma_util_chip'] * 256.0 / cus_of(kname, v), 4)
                if v.get('mfma_util_of_nominal_peak') is not None:
                    e['mfma_busy_of_nominal_peak'] = v['mfma_util_of_nominal_peak']
            pmc_groups[gname] = e
        # HBM bytes of one step = sum over all profiled kernels of calls x bytes / profiled steps (one rowmax launch per step)
        tot = sum(v.get('calls', 0) * (v.get('hbm_read_MB_x2', v.get('hbm_read_MB_per_launch', 0.0)) + v.get('hbm_write_MB_per_launch', 0.0))
                  for v in pmc.values())
        step_traffic = {'GB': round(tot / steps_profiled / 1e3, 3), 'steps_profiled': steps_profiled, 'source': src}
        d = pmc_groups.get(dom_name)
        if d:
            roofline['traffic'] = {'read_MB': d['hbm_read_MB'], 'write_MB': d['hbm_write_MB'], 'per': 'launch', 'hbm_GBps': d['hbm_GBps'],
                                   'source': src + ' -- FETCH_SIZE doubled per the guide for the wide streaming reads'}
            # the same fraction on the profiler's clock: rocprofv3's average duration of the kernel in the committed run
            roofline['frac_hip_events'] = roofline['frac']
            roofline['frac_rocprof'] = round(dom['gflop'] / dom['n'] / d['rocprof_avg_launch_ms'] / peak_of(dom_name), 4)
            roofline['rocprof_avg_launch_ms'] = d['rocprof_avg_launch_ms']
            if 'mfma_busy_chip' in d:
                roofline['mfma_busy_chip'] = d['mfma_busy_chip']
                roofline['mfma_busy_on_its_CUs'] = d['mfma_busy_on_its_CUs']
            if sclk:
                roofline['profile_sclk_mhz'] = sclk
                roofline['profile_sclk_note'] = ('shader clock under this load in the profile named in traffic.source (rocm-smi, power-capped; nominal 2400 MHz): '
                                                 'busy fractions are of actual cycles, x sclk/2400 gives them against the nominal peak')
    except Exception:
        pass
    roofline['step_traffic'] = step_traffic
    for k, v in groups.items():
        v['pmc'] = pmc_groups.get(k)
    return roofline, launches, {k: {'ms': round(v['ms'], 3), 'tflops': round(v['gflop'] / v['ms'], 1) if v['ms'] > 0 else 0,
                                    'frac_of_peak': round(v['gflop'] / v['ms'] / peak_of(k), 4) if v['ms'] > 0 else 0,
                                    'rocprof': v.get('pmc')}
                                for k, v in groups.items()}


def self_profile(args):
    """
    VERDICT r5 item 6: the rocprofv3 figures of the line measured in THIS invocation.  After the timed region rank 0 runs its own
    command again for 6 steps under rocprofv3 -- one --kernel-trace --stats pass and three --pmc passes (FETCH_SIZE, WRITE_SIZE,
    SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; counters never together with a trace, MI355X_MICROARCH.md) -- and condenses them with
    tools/summarize_pmc.py.  Returns {'summary', 'by_grid', 'command', 'seconds'} or {'why': reason} (no rocprofv3 on PATH, already
    under a profiler, a pass failed or timed out): the caller then falls back to the committed summary and says so in `source`.
    """
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    t_start = time.perf_counter()
    exe = shutil.which('rocprofv3')
    if exe is None:
        return {'why': 'rocprofv3 is not on PATH'}
    if any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTRACER')) for k in os.environ):
        return {'why': 'this process already runs under a profiler'}
    tmp = tempfile.mkdtemp(prefix='krk_selfprof_', dir='/tmp')
    cmd = [sys.executable, os.path.abspath(__file__), '--precision', args.precision, '--slots', str(args.slots), '--batch', str(args.batch),
           '--width', str(args.width), '--data', args.data, '--steps', '6', '--warmup', '2', '--preheat-ms', '0', '--no-cpu-baseline',
           '--no-self-profile', '--no-config4-check']
    env = dict(os.environ, TMPDIR='/tmp')
    passes = [('stats', ['--kernel-trace', '--stats']), ('fetch', ['--pmc', 'FETCH_SIZE']), ('write', ['--pmc', 'WRITE_SIZE']),
              ('mfma', ['--pmc', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'])]
    try:
        for tag, flags in passes:
            r = subprocess.run([exe, *flags, '--output-format', 'csv', '-d', os.path.join(tmp, tag), '--', *cmd], cwd='/tmp', env=env,
                               capture_output=True, text=True, timeout=120)          # (a pass takes 3-4 s; a profiler that hangs must not hold the line up)
            if r.returncode:
                return {'why': f'rocprofv3 pass "{tag}" exited {r.returncode}: {(r.stderr or r.stdout)[-200:]!r}'}

        def newest(tag, suffix):
            hits = glob.glob(os.path.join(tmp, tag, '**', '*' + suffix), recursive=True)
            return max(hits, key=os.path.getmtime) if hits else None
        files = [newest('stats', 'kernel_stats.csv'), newest('fetch', 'counter_collection.csv'), newest('write', 'counter_collection.csv'),
                 newest('mfma', 'counter_collection.csv')]
        if not all(files):
            return {'why': 'a rocprofv3 pass wrote no csv'}
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'summarize_pmc.py'), 'self', *files], capture_output=True, text=True,
                           timeout=60)
        if r.returncode:
            return {'why': 'tools/summarize_pmc.py failed: ' + r.stderr[-200:]}
        summary = json.loads(r.stdout)
        # per (kernel, grid size): the launches of one kernel that belong to different launch groups (gemm_x3: projections / linear)
        by_grid, steps = {}, 6
        trace = newest('stats', 'kernel_trace.csv')
        if trace:
            acc = {}
            for row in csv.DictReader(open(trace)):
                m = re.search(r'(\w+_kernel)', row.get('Kernel_Name', ''))
                if not m:
                    continue
                grid = int(row.get('Grid_Size_X', row.get('Grid_Size', 0)) or 0)
                a = acc.setdefault((m.group(1), grid), [0, 0.0])
                a[0] += 1
                a[1] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
            steps = max(1, max((n for (k, _), (n, _) in acc.items() if k == 'rowmax_rows_kernel'), default=6))
            for (k, grid), (n, ns) in acc.items():
                by_grid.setdefault(k, []).append({'grid': grid, 'calls': n, 'per_step': n // steps if n % steps == 0 else -1,
                                                  'avg_us': round(ns / n / 1e3, 1)})
        return {'summary': summary, 'by_grid': by_grid, 'command': 'bench.py ' + ' '.join(cmd[2:]),
                'seconds': round(time.perf_counter() - t_start, 1)}
    except subprocess.TimeoutExpired as e:
        return {'why': f'a rocprofv3 pass did not finish in {e.timeout:.0f} s'}
    except Exception as e:                               # the bench line must come out whatever the profiler does
        return {'why': f'{type(e).__name__}: {e}'[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def config4_golden_check(model, precision):
    """
    BASELINE config 4's 1024 DISTINCT ragged lines (tests/golden/bench_lines.npz: cfg4, kraken's batch-1 strings and top-2 margins)
    through the plan that was just timed, width-bucketed batches of 128 with masked padding: how many strings are identical, and
    whether every differing line lies inside the plan's tie window (4 x its measured logit error).  Outside the timed region.
    """
    try:
        z = np.load(os.path.join(ROOT, 'tests', 'golden', 'bench_lines.npz'), allow_pickle=False)
        want, margin, widths = json.loads(str(z['cfg4_strings'])), z['cfg4_margin'], z['cfg4_widths'].tolist()
    except Exception:
        return None
    tie = 1e-5 if precision == 'f32' else 1e-4
    got = []
    for lo in range(0, len(widths), 128):
        ws = widths[lo:lo + 128]
        xb = torch.zeros(len(ws), 1, 48, max(ws))
        for i, w in enumerate(ws):
            xb[i, ..., :w] = torch.rand(1, 1, 48, w, generator=torch.Generator().manual_seed(50000 + lo + i))[0]
        b, _, _, _ = model.nn.recognize(xb.cuda(), torch.tensor(ws))
        got += model.codec.decode_strings(b)
    diff = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    return {'lines': len(want), 'identical': len(want) - len(diff), 'differ_inside_tie_window': sum(1 for i in diff if margin[i] < tie),
            'differ_outside_tie_window': sum(1 for i in diff if margin[i] >= tie), 'tie_window': tie,
            'source': 'tests/golden/bench_lines.npz: cfg4_strings / cfg4_margin, made by the unmodified reference at batch 1; here '
                      'width-sorted batches of 128, masked padding, after the timed region'}


class _StubEngine:
    """Host stand-in for RecognitionEngine (--stub-engine: plumbing tests without a GPU).  Every line decodes to its rank-tagged index."""

    def __init__(self, rank, slots):
        self.slots, self.q, self.rank, self.k = [None] * slots, [], rank, 0

    def free_slots(self):
        return len(self.slots) - len(self.q)

    def submit(self, x, lens=None):
        assert self.free_slots() > 0
        self.q.append(int(x.shape[0]))

    def collect(self):
        from kraken_amd.vgsl import DecodedBatch
        n = self.q.pop(0)
        lab = (np.arange(n, dtype=np.int32)[:, None] + self.k) % 250 + 1
        self.k += n
        return (DecodedBatch(np.repeat(lab, 2, 1), np.zeros((n, 2), np.int32), np.ones((n, 2), np.int32),
                             np.full((n, 2), 0.5, np.float32), np.full(n, 1 + self.rank % 2, np.int32)), np.full(n, 150, np.int32))

    def set_profiling(self, on):
        pass

    def layer_times(self):
        return []

    def close(self):
        pass


def mode_engine(args, model, rank, world, local_rank, use_dist, kdist):
    stub = args.stub_engine
    dev = torch.device('cpu') if stub else torch.device(f'cuda:{local_rank}')
    N, W = args.batch, args.width
    # resident in HBM before timing; rotated.  Batch 0 of rank 0 is synth_lines(N, W, 1234): the tensor the CPU leg samples and,
    # for --data noise at 256 x 1200, the one tests/golden/bench_lines.npz pins line by line against kraken
    xs = [synth_lines(N, W, 1234 + rank + 1000 * b, args.data).to(dev) for b in range(1 if stub else 4)]
    if args.data == 'dense' and not stub:
        densify(model, xs[0][:64])
    # the product's sharded recogniser (kraken_amd/dist.py): one pipelined engine per rank, one gather of decoded tuples
    sr = kdist.ShardedRecognizer(model, device=local_rank, batch=N, slots=args.slots, max_width=W,
                                 engine_factory=(lambda: _StubEngine(rank, args.slots)) if stub else None)
    engine = sr.engine
    if args.host_input:
        xs = [x.cpu().pin_memory() for x in xs]
    codec = model.codec
    n_chars, host_s, first = [0], [0.0], []

    def to_text(decoded, olens):
        t0 = time.perf_counter()
        strings = codec.decode_strings(decoded)        # host codec: label tuples -> text, inside the timed region
        host_s[0] += time.perf_counter() - t0
        n_chars[0] += sum(map(len, strings))
        if not first:
            first.append(strings)                      # the first batch of the timed region (= xs[0]): checked against the CPU leg

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        if not stub:
            torch.cuda.synchronize()

    # untimed, before the W warm-up steps: bring the clock governor (and the allocator, and RCCL below) to where a running service is
    # ... in chunks of 16 steps, for at least --preheat-ms, and on until two consecutive chunks run at the same rate (2 %) or eight times
    # that long: on some boxes 400 ms is not enough after an idle period (profiles/r06_preheat_adaptive.txt: the same command gave
    # 81 k and 118 k lines/s minutes apart on one box, the kernels' own times identical)
    preheat_steps, t_pre, rates = 0, time.perf_counter(), []
    while not stub and args.preheat_ms > 0:
        t_c = time.perf_counter()
        sr.stream((xs[i % len(xs)] for i in range(16)), to_text)
        torch.cuda.synchronize()
        rates.append(16 / (time.perf_counter() - t_c))
        preheat_steps += 16
        spent = (time.perf_counter() - t_pre) * 1e3
        steady = len(rates) >= 2 and abs(rates[-1] - rates[-2]) <= 0.02 * rates[-1]
        if (spent >= args.preheat_ms and steady) or spent >= 8 * args.preheat_ms:
            break
    done = sr.stream((xs[i % len(xs)] for i in range(args.warmup)), to_text)
    if use_dist:
        sr.gather(done[-2:], force=args.force_dist)      # untimed: RCCL builds its communicator on first use
    engine.set_profiling(True)
    n_chars[0], host_s[0] = 0, 0.0
    first.clear()
    barrier()
    t0 = time.perf_counter()
    done = sr.stream((xs[i % len(xs)] for i in range(args.steps)), to_text)
    # every line this rank decoded in the timed region travels in one exchange (RCCL all_gather of compact tuples)
    gathered = sr.gather(done, force=args.force_dist) if use_dist else None
    gathered_lines = sum(len(b.counts) for b in gathered) if use_dist else sum(len(b.counts) for b, _ in done)
    barrier()
    dt = time.perf_counter() - t0
    gather_ms = sr.gather_ms if use_dist else 0.0
    per_rank = None
    if use_dist:
        # every rank's own clock and gather time (tools/scale_preflight.sh logs them per rank), then the MAX over ranks for the line
        mine = torch.tensor([dt, gather_ms, 1e6 * host_s[0] / max(1, N * args.steps)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(every, mine)
        per_rank = {'lines_per_s': [round(N * args.steps / float(e[0].item()), 1) for e in every],
                    'gather_ms': [round(float(e[1].item()), 3) for e in every],
                    # each rank's own host work per line it decoded (codec: tuples -> strings): must not grow with the number of ranks
                    'host_us_per_line': [round(float(e[2].item()), 3) for e in every]}
        t = mine.clone()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, gather_ms = float(t[0].item()), float(t[1].item())
    lines = N * args.steps * world
    value = lines / dt
    gathered_lines = int(gathered_lines)
    assert gathered_lines == lines, (gathered_lines, lines)
    ranks_seen = torch.distributed.get_world_size() if use_dist else 1
    assert ranks_seen == world, (ranks_seen, world)
    out = {
        'metric': METRIC, 'value': round(value, 1), 'unit': 'lines/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': DTYPES[args.precision],
        'data': 'synthetic' + {'noise': '', 'strokes': ' (strokes: 5 % of the columns U(0.5,1), the rest 0)', 'dense': ' (glyph cells; LSTM input projections x10 and OUTPUT biases recalibrated so that a line decodes at real text density)'}[args.data] + (' (pinned host input per step: PCIe-inclusive)' if args.host_input else ''),
        'config': {'workload': f'BENCH-A VGSL recogniser (3.17M params, random init seed 0), {N} lines 1x48x{W} per GPU '
                               f'per step, greedy CTC decode, label tuples to host, host codec to strings',
                   'inputs': 'pinned host tensors, copied per step (PCIe-inclusive)' if args.host_input else 'resident in HBM before the timed region',
                   'lines_per_gpu_step': N, 'width': W, 'input_set': args.data, 'slots': args.slots, 'precision': args.precision,
                   'parallelism': f'dp{world}', 'whole_path_tflops': round(value * 2.778e-3 * (W / 1200.0), 2)},
        'ranks_in_collective': ranks_seen, 'collective_backend': torch.distributed.get_backend() if use_dist else None,
        'gather_ms': round(gather_ms, 3), 'gathered_lines': gathered_lines, 'decoded_chars': int(n_chars[0]),
        'chars_per_line': round(n_chars[0] / max(1, N * args.steps), 2),
        'host_us_per_line': {'codec_strings': round(1e6 * host_s[0] / max(1, N * args.steps), 3)},
    }
    if per_rank:
        out['per_rank'] = per_rank
    out['warmup_effective'] = args.warmup + preheat_steps      # every untimed step in front of the timed region
    out['preheat'] = {'steps': preheat_steps, 'ms': args.preheat_ms, 'chunk_rates_steps_per_s': [round(r, 1) for r in rates[-6:]],
                      'note': 'untimed steps in front of the W warm-up steps (clock governor out of its idle state); --preheat-ms 0 switches it off'}
    out['_first_strings'] = first[0] if first else []
    if not stub and done:
        # the host side at this text density, outside the timed region: label tuples -> LineResult (text + cut positions +
        # confidences per line: what the record assembly of rpred consumes, kraken lib/codec.py:148-195 + rpred.py:226-250)
        from kraken_amd.rpred import _decode_lines
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            for decoded, olens in done[:8]:
                _decode_lines(codec, decoded, olens)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out['host_us_per_line']['line_results'] = round(1e6 * best / (N * len(done[:8])), 3)
    if stub:
        out['data'] = 'STUB ENGINE -- plumbing test without a GPU, no device work: `value` is meaningless'
        out['dtype'] = 'none (stub)'
        return out
    fresh = None
    if rank == 0 and world == 1 and not args.no_self_profile:
        fresh = self_profile(args)           # (the parent keeps its engine: it is idle, and 288 GB hold both)
    elif rank == 0:
        fresh = {'why': '--no-self-profile' if args.no_self_profile else 'multi-rank run'}
    roofline, launches, groups = roofline_of(engine, args.precision, fresh)
    if fresh and fresh.get('seconds'):
        roofline['self_profile_s'] = fresh['seconds']
    out.update({
        'roofline': roofline,
        'launches': [{'name': l['name'], 'ms': round(l['ms'], 3), 'tflops': round(l['gflop'] / l['ms'], 1) if l['ms'] > 0 else 0}
                     for l in launches],
        'groups': groups})
    return out


def _page_of_lines(n, w, h, mode, seed=7):
    """One synthetic page: n text-line boxes of w x h stacked vertically (random ink, so no line is flat)."""
    from PIL import Image
    from kraken_amd.containers import BBoxLine, Segmentation
    rng = np.random.default_rng(seed)
    shape = (n * h, w) if mode == 'L' else (n * h, w, 3)
    page = Image.fromarray(rng.integers(0, 256, shape, dtype=np.uint8), mode)
    seg = Segmentation(type='bbox', imagename='synthetic', text_direction='horizontal-lr', script_detection=False,
                       lines=[BBoxLine(id=f'l{i}', bbox=[0, i * h, w, (i + 1) * h]) for i in range(n)])
    return page, seg


def mode_api(args, rank, local_rank):
    """Lines/s through the legacy generator API: 1-channel bbox lines (CenterNormalizer dewarp) and RGB lines (rectangular crops), each
    prepared on the device and, for comparison, with scipy / PIL on the host."""
    import warnings
    import kraken_amd
    from kraken_amd import rpred as R
    from kraken_amd.engine import RecognitionEngine
    from kraken_amd.models import TorchSeqRecognizer
    from kraken_amd.specs import BENCH_A, BENCH_A_RGB, bench_codec
    out = {}
    n, W = args.api_lines, args.width
    if os.environ.get('KRK_API_NOGC'):               # probe: how much of the host time is the cyclic collector?
        import gc
        gc.disable()
    for name, spec, mode in (('bbox_L_dewarped_on_device', BENCH_A, 'L'), ('bbox_RGB_prepared_on_device', BENCH_A_RGB, 'RGB')):
        torch.manual_seed(0)
        m = kraken_amd.TorchVGSLModel(vgsl=spec, codec=bench_codec())
        m.seg_type, m.model_type = 'bbox', ['recognition']
        m.to(f'cuda:{local_rank}')
        try:
            m.nn.set_precision(args.precision)
            m.nn.plan(local_rank)
        except Exception:
            m.nn.set_precision('f32')               # 3 input channels: the split-bf16 first convolution takes 1 channel
        net = TorchSeqRecognizer(m, device=f'cuda:{local_rank}')
        # line crops W-32 wide and 48 high: the network input is 48 x W after the 16 px padding
        page, seg = _page_of_lines(n, W - 32, 48, mode)
        res = {}
        if os.environ.get('KRK_PROFILE_API'):          # where the host time of the API path goes (stderr)
            import cProfile
            import pstats
            pr = cProfile.Profile()
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                list(R.rpred(net, page, seg, bidi_reordering=False, num_line_workers=args.api_workers))
                pr.enable()
                list(R.rpred(net, page, seg, bidi_reordering=False, num_line_workers=args.api_workers))
                pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats('tottime').print_stats(14)
        for label, dev_prep in (('api', True), ('api_host_preparation', False)):
            R.DEVICE_PREP = dev_prep
            best, reps = 0.0, []
            # the first pass creates plans, pinned buffers and the allocator's blocks; a pass is 50..100 ms of wall time, so single
            # passes scatter (collector pauses, thread start-up): the MEDIAN of the warm passes is the number reported (what a caller of
            # rpred sees), the best pass next to it; both preparations get the same eight passes and the same collector treatment
            for rep in range(8):
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    t0 = time.perf_counter()
                    # host-prepared lines (PIL resize / scipy dewarp, mostly outside the GIL) take more threads than the
                    # device-prepared case, whose only host work is the GIL-bound page conversion
                    workers = args.api_workers if (dev_prep and mode == 'RGB') else max(16, args.api_workers)
                    recs = list(R.rpred(net, page, seg, bidi_reordering=False, num_line_workers=workers))
                    dt = time.perf_counter() - t0
                assert len(recs) == n and all(r.prediction for r in recs)
                best = max(best, n / dt)
                reps.append(round(n / dt, 1))
                if rep == 0 and not os.environ.get('KRK_API_NOFREEZE'):
                    # what a long-running service does once its models are loaded (gc.freeze: everything allocated so far leaves
                    # the collector's generations): without it every other 60 ms pass pays a ~55 ms full collection of the
                    # interpreter's heap (torch's modules), triggered by the ~15 container objects a record consists of
                    import gc
                    del recs
                    gc.collect()
                    gc.freeze()
                    res['gc'] = 'gc.freeze() after the first pass'
            res[label + '_lines_per_s'] = round(float(np.median(reps[2:])), 1)
            res[label + '_best_pass'] = round(best, 1)
            res[label + '_all_passes'] = reps
        R.DEVICE_PREP = True
        # the same model with inputs resident in HBM (what the default mode measures)
        eng = RecognitionEngine(m, device=local_rank, max_batch=256, max_width=W, slots=args.slots)
        x = torch.rand(256, m.input[1], 48, W, device=f'cuda:{local_rank}')
        for _ in range(3):
            eng.submit(x)
            eng.collect()
        steps = max(n // 256, 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            if eng.free_slots() == 0:
                m.codec.decode_strings(eng.collect()[0])
            eng.submit(x)
        while eng.free_slots() < len(eng.slots):
            m.codec.decode_strings(eng.collect()[0])
        torch.cuda.synchronize()
        res['engine_resident_input_lines_per_s'] = round(256 * steps / (time.perf_counter() - t0), 1)
        res['plan'] = 'bf16x3' if m.nn.precision == 2 else 'f32'
        res['api_over_engine'] = round(res['api_lines_per_s'] / res['engine_resident_input_lines_per_s'], 3)
        eng.close()
        out[name] = res
    head = out['bbox_RGB_prepared_on_device']
    return {'metric': 'text lines/sec through the reference API (kraken_amd.rpred.rpred generator: crop, transform, network, records)',
            'value': head['api_lines_per_s'], 'unit': 'lines/s', 'n_gpus': 1, 'steps': 1, 'warmup': 0,
            'ms_per_step': round(1e3 * n / head['api_lines_per_s'], 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': head['plan'], 'data': 'synthetic',
            'config': {'workload': f'one synthetic page of {n} bbox lines {W - 32}x48 through rpred() -> ocr_records; value = the RGB '
                                   f'(device-prepared) case, `cases` holds both', 'host_threads': args.api_workers},
            'cases': out}


def mode_config4(args, model, local_rank):
    """
    BASELINE config 4: 1024 lines, widths U{400..2400}, length bucketing + packed LSTM.  Every timed step is a FRESH set of 1024
    widths (seed 40 + step), bucketed by the product's rule (kraken_amd.rpred.width_buckets: width-sorted, at most --bucket-lines
    lines and --bucket-px padded pixels per device batch), buckets resident in HBM, --slots buckets in flight; `value` = median over
    the steps.  `buckets`: one synchronous, profiled pass over step 0's buckets -- padding share, time alone, the recurrent launches'
    share and T_max -- so that the distance to the uniform-width headline is accounted for.  `pcie_inclusive`: the same lines as
    uint8 host images through the API's LinePipeline (pinned staging, bucketing on the fly).
    """
    from kraken_amd import rpred as R
    from kraken_amd.engine import RecognitionEngine
    from kraken_amd.models import TorchSeqRecognizer
    dev = f'cuda:{local_rank}'
    steps = max(args.steps if args.steps != 50 else 20, 1)
    max_lines, px = args.bucket_lines, args.bucket_px

    def make_step(k):
        widths = np.random.RandomState(40 + k).randint(400, 2401, size=1024)
        out = []
        for idx in R.width_buckets(widths, max_lines, px):
            ws = widths[idx].astype(np.int32)
            g = torch.Generator(device=dev).manual_seed(4100 + k)
            x = torch.rand(len(idx), 1, 48, int(ws.max()), generator=g, device=dev)
            x *= (torch.arange(int(ws.max()), device=dev)[None, :] < torch.from_numpy(ws).to(dev)[:, None])[:, None, None, :]
            out.append((x, ws))
        return widths, out

    sets = [make_step(k) for k in range(steps + 1)]              # set `steps` = the warm-up pass (buffers grow to their final size)
    cap = max(len(ws) for _, bs in sets for _, ws in bs)
    eng = RecognitionEngine(model, device=local_rank, max_batch=cap, max_width=2400, slots=args.slots)

    def run(buckets):
        n_out = 0
        for x, ws in buckets:
            if eng.free_slots() == 0:
                n_out += len(model.codec.decode_strings(eng.collect()[0]))
            eng.submit(x, ws)
        while eng.free_slots() < len(eng.slots):
            n_out += len(model.codec.decode_strings(eng.collect()[0]))
        assert n_out == 1024
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(sets[steps][1])
    torch.cuda.synchronize()
    first_pass = time.perf_counter() - t0
    run(sets[steps][1])
    times = []
    for k in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(sets[k][1])
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    # the same steps back to back (a stream of pages: the next job's first buckets are submitted while the last ones of this job drain)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_sub = 0
    for k in range(steps):
        for x, ws in sets[k][1]:
            if eng.free_slots() == 0:
                model.codec.decode_strings(eng.collect()[0])
            eng.submit(x, ws)
            n_sub += len(ws)
    while eng.free_slots() < len(eng.slots):
        model.codec.decode_strings(eng.collect()[0])
    torch.cuda.synchronize()
    stream_dt = time.perf_counter() - t0
    assert n_sub == 1024 * steps
    px_total = [float(np.sum(w)) for w, _ in sets[:steps]]
    eq = float(np.median([p / 1200.0 / t for p, t in zip(px_total, times)]))
    # the breakdown: step 0's buckets one at a time (nothing else in flight), profiled
    eng.set_profiling(True)
    rows, alone_sum = [], 0.0
    for x, ws in sets[0][1]:
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.submit(x, ws)
            eng.collect()
            dt = time.perf_counter() - t0
        import ctypes
        h = eng.last_slot.plan.handle
        nst = eng.lib.krk_plan_num_steps(h)
        ms = (ctypes.c_float * nst)()
        grp = {}
        if eng.lib.krk_plan_layer_ms(h, ms, nst) >= 0:
            for i in range(nst):
                name = eng.lib.krk_plan_layer_name(h, i).decode()
                grp[name] = grp.get(name, 0.0) + float(ms[i])
        rec = sum(v for k_, v in grp.items() if k_.startswith('lstm_rec'))
        n, wmax = len(ws), int(ws.max())
        alone_sum += dt
        rows.append({'lines': n, 'w_min': int(ws.min()), 'w_max': wmax, 'padding_share': round(1.0 - float(ws.sum()) / (n * wmax), 4),
                     'ms_alone': round(1e3 * dt, 3), 'ms_kernels': round(sum(grp.values()), 3), 'ms_recurrent': round(rec, 3),
                     'T_max': wmax // 8, 'us_per_recurrent_step': round(1e3 * rec / max(3 * (wmax // 8), 1), 3),
                     'padded_Mpx': round(n * wmax * 48 / 1e6, 2)})
    eng.set_profiling(False)
    eng.close()
    padded = sum(r['lines'] * r['w_max'] for r in rows)
    account = {'lines_per_s_of_the_uniform_headline': 'see the default mode of the same build (256 x 1200)',
               'true_px_per_step0': int(px_total[0]), 'padded_px_per_step0': int(padded), 'padding_share_step0': round(1.0 - px_total[0] / padded, 4),
               'sum_ms_alone_step0': round(1e3 * alone_sum, 3), 'ms_pipelined_step0': round(1e3 * times[0], 3),
               'first_pass_ms (plans and buffers grow)': round(1e3 * first_pass, 3)}
    # (b) the same kind of lines as uint8 line IMAGES on the host (what a line extractor hands over), through the API's pipeline:
    # packed 1 byte per pixel over PCIe, padded / scaled / inverted on the device (krk_prep_crops), bucketed on the fly
    pcie = None
    if not args.no_pcie:
        net = TorchSeqRecognizer(model, device=dev)
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(8)
        widths = sets[0][0]
        rng = np.random.default_rng(41)
        crops = [rng.integers(0, 256, (48, int(w) - 32), dtype=np.uint8) for w in widths]     # the 16 px padding is added on the device
        best = None
        for rep in range(3):
            pipe = R.LinePipeline(net, batch_size=max_lines, pool=pool)
            t0 = time.perf_counter()
            pipe.submit_crops(list(enumerate(crops)), 16)
            got = {}
            while pipe.pending():
                got.update(pipe.drain(block=True))
            got.update(pipe.drain())
            dt = time.perf_counter() - t0
            assert len(got) == 1024 and all(got[i].out_width == int(widths[i]) // 8 for i in range(1024))
            best = dt if best is None or rep == 0 else min(best, dt)
            pipe.close()
        pcie = {'value': round(1024 / best, 1), 'unit': 'lines/s',
                'note': f'uint8 line images on the host -> LinePipeline.submit_crops (bucketing, packed pinned staging at 1 B/px, '
                        f'padding / scaling / inversion on the device, {R.ENGINE_SLOTS} batches in flight)'}
    return {'metric': 'text lines/sec, BASELINE config 4 (1024 lines, W ~ U{400..2400}, length bucketing + packed LSTM)',
            'value': round(1024 / med, 1), 'unit': 'lines/s', 'n_gpus': 1, 'steps': steps, 'warmup': 2,
            'ms_per_step': round(1e3 * med, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': DTYPES[args.precision], 'data': 'synthetic',
            'config': {'workload': f'{steps} steps of 1024 fresh lines 1x48xW, W ~ U{{400..2400}} (seed 40 + step), width-sorted into buckets of '
                                   f'<= {max_lines} lines' + (f' and <= {px} padded px' if px else '') + f', {args.slots} buckets in flight, '
                                   f'buckets resident in HBM; value = median step', 'bucket_lines': max_lines, 'bucket_px': px,
                       'buckets_per_step': round(float(np.mean([len(b) for _, b in sets[:steps]])), 1),
                       'mean_width': round(float(np.mean(px_total)) / 1024, 1),
                       'equivalent_1200px_lines_per_s': round(eq, 1)},
            'step_ms': {'median': round(1e3 * med, 3), 'min': round(1e3 * min(times), 3), 'max': round(1e3 * max(times), 3)},
            'back_to_back': {'lines_per_s': round(1024 * steps / stream_dt, 1),
                             'equivalent_1200px_lines_per_s': round(float(np.sum(px_total)) / 1200.0 / stream_dt, 1),
                             'note': f'the {steps} jobs submitted as one stream (no drain between jobs)'},
            'buckets': rows, 'account': account, 'pcie_inclusive': pcie}


def share_device_product_check(model, rank, world):
    """
    --share-device: the product call on the same box.  `ShardedRecognizer.recognize_lines` over a ragged set of lines (every rank
    holds the same list) must return one result per line in INPUT order on every rank, identical to what a single unsharded
    engine returns for the same lines.
    """
    from kraken_amd import dist as kdist
    g = torch.Generator().manual_seed(99)
    widths = torch.randint(200, 1201, (97,), generator=g).tolist()
    lines = [torch.rand(1, 48, w, generator=g) for w in widths]
    sr = kdist.ShardedRecognizer(model, device=0, batch=32, slots=2, max_width=1200)
    got = sr.recognize_lines(lines)
    sr.close()
    # the unsharded answer: the same call in a one-rank group (every rank computes it for itself)
    groups = [torch.distributed.new_group([r]) for r in range(world)]       # collective: every rank creates every group
    solo = kdist.ShardedRecognizer(model, device=0, batch=32, slots=2, max_width=1200, group=groups[rank])
    want = solo.recognize_lines(lines)
    solo.close()
    same = [a.text == b.text and list(a.starts) == list(b.starts) and list(a.ends) == list(b.ends) and a.out_width == b.out_width
            for a, b in zip(got, want)]
    ok = torch.tensor([int(all(same) and len(got) == len(lines))])
    torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
    return {'lines': len(lines), 'ranks': world, 'results_in_input_order_and_identical_to_one_rank': bool(ok.item()),
            'nonempty': sum(bool(r.text) for r in got)}


def golden_strings(args):
    """kraken's own strings for the first timed batch of rank 0, when the run is the pinned configuration (BASELINE config 2)."""
    if args.data != 'noise' or args.batch != 256 or args.width != 1200:
        return None
    try:
        z = np.load(os.path.join(ROOT, 'tests', 'golden', 'bench_lines.npz'), allow_pickle=False)
        return json.loads(str(z['cfg2_strings']))
    except Exception:
        return None


def launch_ranks(args) -> int:
    """
    `python bench.py --gpus N` without a launcher: become the launcher.  Spawns N copies of this command, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1), rank 0 prints the JSON line.
    Refuses (exit 3) when fewer than N devices are visible -- never a silent 1-GPU run labelled as N.
    """
    import socket
    import subprocess
    n = args.gpus
    if not args.stub_engine:
        from kraken_amd import _lib
        have = _lib.device_count()
        need = 1 if args.share_device else n
        if have < need:
            print(f'bench.py: --gpus {n} but only {have} HIP device(s) visible; refusing to run', file=sys.stderr)
            return 3
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            rc = p.wait() or rc
            if rc:
                break
    finally:
        for p in procs:                    # a rank that died must not leave its peers waiting in a collective
            if p.poll() is None:
                p.terminate()
    return rc


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.share_device and (args.mode != 'engine' or args.stub_engine):
        raise SystemExit('--share-device is a plumbing run of the default mode on a real device')
    import kraken_amd
    from kraken_amd import _lib, dist as kdist
    # rank r keeps the CPUs of its GPU's NUMA node, shared with the other ranks on that node (kraken_amd/dist.py: rank_cpu_block)
    cpus_kept = (kdist.pin_rank_to_cpus(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world))) if world > 1
                 else {'cpus': os.cpu_count() or 1, 'numa_node': None, 'first_cpu': 0})
    if args.share_device:
        local_rank = 0                      # every rank on HIP device 0
    from kraken_amd.specs import BENCH_A, bench_codec

    stub = args.stub_engine
    if stub and args.mode != 'engine':
        raise SystemExit('--stub-engine is a plumbing test of the default mode only')
    if not stub:
        _lib.require_gpu()
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f'rank {rank}: LOCAL_RANK={local_rank} but {torch.cuda.device_count()} device(s) visible')
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        kdist.init(backend='gloo' if (stub or args.share_device) else 'nccl')

    torch.manual_seed(0)
    model = kraken_amd.TorchVGSLModel(vgsl=BENCH_A, codec=bench_codec())
    if not stub:
        model.to(torch.device(f'cuda:{local_rank}'))
        model.nn.set_precision(args.precision)
    if args.mode == 'api':
        out = mode_api(args, rank, local_rank)
    elif args.mode == 'config4':
        out = mode_config4(args, model, local_rank)
    else:
        out = mode_engine(args, model, rank, world, local_rank, use_dist, kdist)
    first_strings = out.pop('_first_strings', None)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == 'engine' and not stub:
        out['cpu_baseline'], cpu_strings = cpu_baseline(model, args.width, args.cpu_lines, args.data, args.batch)
        # the number above is tied to parity: the strings of the first timed batch against the CPU leg's strings of the same lines
        k = min(len(cpu_strings), len(first_strings or []))
        out['parity_checked'] = {'lines': k, 'identical': sum(a == b for a, b in zip(first_strings[:k], cpu_strings[:k])),
                                 'against': f"cpu_baseline leg ({out['cpu_baseline']['kind']}), same lines, same weights"}
    if rank == 0 and first_strings and args.mode == 'engine' and not stub:
        g = golden_strings(args)
        if g is not None:
            out.setdefault('parity_checked', {})['kraken_golden'] = {
                'lines': len(g), 'identical': sum(a == b for a, b in zip(first_strings, g)),
                'source': 'tests/golden/bench_lines.npz: cfg2_strings, made by the unmodified reference (tests/golden/make_golden.py)'}
    if rank == 0 and world == 1 and args.mode == 'engine' and not stub and not args.no_config4_check and golden_strings(args) is not None \
            and args.precision in ('f32', 'bf16x3'):
        c4 = config4_golden_check(model, args.precision)
        if c4 is not None:
            out.setdefault('parity_checked', {})['kraken_golden_config4'] = c4
    if world > 1:
        out['host_cpus_per_rank'] = cpus_kept
    if args.share_device:
        out['recognize_lines_check'] = share_device_product_check(model, rank, world)
        out['n_gpus'] = 1
        out['ranks_on_one_device'] = world
        out['metric'] = 'PLUMBING RUN, NOT A SCALING NUMBER -- ' + out['metric']
        out['data'] += f'; {world} ranks (processes) share HIP device 0, gloo collective: what this line shows is that two real engines, ' \
                       'ShardedRecognizer.stream + gather and recognize_lines work across processes on hardware; its value says nothing about scaling'
        out['scaling'] = 'none (one device)'
    if use_dist:
        torch.distributed.destroy_process_group()
    if rank == 0:
        try:                                           # RCCL / the HIP runtime print through C stdio: flush it first so that
            import ctypes                              # the JSON line is the LAST line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
