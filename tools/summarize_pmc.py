#!/usr/bin/env python
"""
Condenses rocprofv3 outputs (kernel stats + separate --pmc passes) into a small JSON summary that
is committed under profiles/.  Usage:

    python tools/summarize_pmc.py <round tag> <kernel_stats.csv> <FETCH_SIZE csv> <WRITE_SIZE csv> <MFMA csv> [clock.json]

(files, not directories: tools/profile_round.sh passes the NEWEST file of each freshly emptied output directory)

Counter conventions (/opt/skills/guides/MI355X_MICROARCH.md, sections HBM + rocprofv3): FETCH_SIZE / WRITE_SIZE are in
KiB per dispatch and are reported RAW.  Calibration against known byte counts in these kernels:
WRITE_SIZE is exact (LSTM output 61.4 MB -> 61.6, conv1 output 472 MB -> 472.1); FETCH_SIZE is exact for
the 4-byte staging reads of the f32 convolutions (315 MB expected incl. halo -> 315.7) but reads HALF
for the LSTM's xproj stream (245.8 MB + weights expected -> 129): the guide's gfx950 caveat ("exactly
1/2 of a wide coalesced streaming read").  `hbm_read_MB_x2` gives the doubled value for the kernels
whose dominant read stream is known to be affected.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs * kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs -- a fraction of the cycles the chip ACTUALLY ran, and the
MI355X clocks to its power budget: `clock` (tools/clock_sample.py: rocm-smi's sclk and package power during an un-profiled run of the
same bench command) says at which clock, `mfma_util_of_nominal_peak` = mfma_util_chip x sclk / 2400 MHz is the same figure against
the 2.5 PFLOP/s the nominal clock would give, and `effective_clock_mhz` = GRBM_GUI_ACTIVE / 8 / the dispatch's wall time in the
counter pass itself (the guide's "effective clock"; counter passes serialise the dispatches, so it is the clock of a kernel alone).
"""
import collections
import csv
import json
import os
import re
import sys

HALVED_READ_KERNELS = ('lstm_f32_kernel', 'lstm_x3_kernel', 'lstm_ws_kernel')   # calibrated: xproj stream reads half


def short(name):
    m = re.search(r'(\w+_kernel(?:<[^>]*>)?)', name)
    return m.group(1).replace(' ', '') if m else name[:40]


def pmc(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
        try:        # wall time of the dispatch in the counter pass (ns), once per dispatch and counter
            d[k]['_ns:' + r['Counter_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        except (KeyError, ValueError):
            pass
    return d


def main():
    tag, stats_file, fetch_dir, write_dir, mfma_dir = sys.argv[1:6]
    clock = None
    if len(sys.argv) > 6 and os.path.exists(sys.argv[6]):
        clock = json.load(open(sys.argv[6]))
        clock.pop('all_samples', None)
    out = {'round': tag, 'sources': {'kernel_stats': os.path.basename(stats_file), 'pmc': [os.path.basename(os.path.dirname(os.path.dirname(p))) or p
                                                                                         for p in (fetch_dir, write_dir, mfma_dir)]},
           'note': 'durations (avg_us) are from the kernel-trace run at the benchmark slot count; counters from the passes named in `sources` '
                   '(solo = --slots 1, load = the benchmark slot count)',
           'kernels': {}}
    sclk = None
    if clock:
        out['clock'] = clock
        sclk = (clock.get('sclk_mhz') or {}).get('median')
    for r in csv.DictReader(open(stats_file)):
        k = short(r['Name'])
        out['kernels'][k] = {'calls': int(r['Calls']), 'avg_us': round(float(r['AverageNs']) / 1e3, 1),
                             'total_ms': round(float(r['TotalDurationNs']) / 1e6, 3), 'pct': float(r['Percentage'])}
    fe, wr, mf = pmc(fetch_dir), pmc(write_dir), pmc(mfma_dir)
    for k, e in out['kernels'].items():
        if k in fe:
            kib = sum(fe[k]['FETCH_SIZE']) / len(fe[k]['FETCH_SIZE'])
            e['hbm_read_MB_per_launch'] = round(kib * 1024 / 1e6, 1)
            if k.startswith(HALVED_READ_KERNELS):
                e['hbm_read_MB_x2'] = round(2 * kib * 1024 / 1e6, 1)
        if k in wr:
            kib = sum(wr[k]['WRITE_SIZE']) / len(wr[k]['WRITE_SIZE'])
            e['hbm_write_MB_per_launch'] = round(kib * 1024 / 1e6, 1)
        if k in mf and 'GRBM_GUI_ACTIVE' in mf[k]:
            busy = sum(mf[k]['SQ_VALU_MFMA_BUSY_CYCLES']) / len(mf[k]['SQ_VALU_MFMA_BUSY_CYCLES'])
            cyc = sum(mf[k]['GRBM_GUI_ACTIVE']) / len(mf[k]['GRBM_GUI_ACTIVE']) / 8.0
            e['mfma_util_chip'] = round(busy / (1024 * cyc), 4) if cyc else None
            if cyc and sclk:
                e['mfma_util_of_nominal_peak'] = round(busy / (1024 * cyc) * sclk / 2400.0, 4)
            ns = mf[k].get('_ns:GRBM_GUI_ACTIVE')
            if cyc and ns and sum(ns) > 0:
                e['effective_clock_mhz'] = round(cyc / (sum(ns) / len(ns)) * 1e3, 1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
