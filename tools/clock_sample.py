#!/usr/bin/env python
"""
Runs a command and samples the shader clock and the package power (rocm-smi) while it runs:

    python tools/clock_sample.py out.json [--period 0.3] -- python bench.py --steps 3000 --no-cpu-baseline

out.json: every sample, and the median / min / max sclk and power of the samples UNDER LOAD (power >= 60 % of the highest power seen:
the start-up of the command -- imports, plan creation -- idles the chip and must not dilute the figure).  The MI355X clocks to its
power budget (MI355X_MICROARCH.md, "DVFS give-back"): MFMA-busy fractions are of ACTUAL cycles, so the clock a profile was taken at
belongs next to them (tools/summarize_pmc.py reads this file).
"""
import json
import re
import statistics
import subprocess
import sys
import threading
import time

NOMINAL_MHZ = 2400          # MI355X_MICROARCH.md: max clock


def parse_smi(text):
    """(sclk MHz, package power W) of GPU 0 from `rocm-smi --showclocks --showpower` (None where a field is missing)."""
    clk = re.search(r'GPU\[0\]\s*:\s*sclk clock level:\s*\S+\s*\((\d+)Mhz\)', text)
    pw = re.search(r'GPU\[0\]\s*:\s*[^\n]*Power \(W\):\s*([\d.]+)', text)
    return (int(clk.group(1)) if clk else None, float(pw.group(1)) if pw else None)


def summarise(samples):
    loaded = [s for s in samples if s['sclk_mhz'] is not None and s['power_w'] is not None]
    if not loaded:
        return {'samples': len(samples), 'under_load': 0}
    top = max(s['power_w'] for s in loaded)
    hot = [s for s in loaded if s['power_w'] >= 0.6 * top]
    clk = [s['sclk_mhz'] for s in hot]
    pw = [s['power_w'] for s in hot]
    return {'samples': len(samples), 'under_load': len(hot), 'rule': 'power >= 60 % of the highest power sampled',
            'sclk_mhz': {'median': statistics.median(clk), 'min': min(clk), 'max': max(clk)},
            'power_w': {'median': statistics.median(pw), 'max': max(pw)},
            'nominal_mhz': NOMINAL_MHZ, 'sclk_over_nominal': round(statistics.median(clk) / NOMINAL_MHZ, 4)}


def main():
    args = sys.argv[1:]
    out = args.pop(0)
    period = 0.3
    if args and args[0] == '--period':
        period = float(args[1])
        args = args[2:]
    assert args and args[0] == '--', __doc__
    cmd = args[1:]
    samples, stop = [], threading.Event()

    def sampler():
        t0 = time.time()
        while not stop.is_set():
            try:
                r = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=10)
                clk, pw = parse_smi(r.stdout)
            except Exception:
                clk = pw = None
            samples.append({'t': round(time.time() - t0, 2), 'sclk_mhz': clk, 'power_w': pw})
            stop.wait(period)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(timeout=15)
    res = summarise(samples)
    res['command'] = ' '.join(cmd)
    res['all_samples'] = samples
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != 'all_samples'}))
    sys.exit(rc)


if __name__ == '__main__':
    main()
