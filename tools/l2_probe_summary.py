"""Condenses the counter passes of tools/l2_probe.sh: per kernel the average of every counter over its launches."""
import collections
import csv
import glob
import re
import sys

res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel(?:<[^>]*>)?)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:30]
        res[k][r['Counter_Name']].append(float(r['Counter_Value']))
        try:
            res[k]['_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        except (KeyError, ValueError):
            pass
for k, v in sorted(res.items()):
    if not any(s in k for s in ('gemm_x3', 'conv_taps', 'conv_x3p', 'lstm_ws', 'conv1_x3')):
        continue
    d = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, f"launches {len(v.get('GRBM_GUI_ACTIVE', v['_ns'])) }", f"avg {d.pop('_ns') / 1e3:.1f} us")
    for c in sorted(d):
        print(f'    {c:36s} {d[c]:16.0f}')
