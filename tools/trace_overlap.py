"""Concurrency histogram of a rocprofv3 kernel trace: how much of the steady-state window has 0/1/2/3+ full-chip
(convolution / projection) kernels and 0..4 recurrent kernels running.  Usage: python tools/trace_overlap.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r['Kernel_Name']
    kind = 'lstm' if 'lstm_' in n else ('small' if ('rowmax' in n or 'collapse' in n or 'copyBuffer' in n) else 'conv')
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), kind))
ev.sort()
ls = [e[0] for e in ev if e[2] == 'lstm']
t0, t1 = ls[int(len(ls) * 0.3)], ls[int(len(ls) * 0.8)]
pts = []
for s, e, k in ev:
    s, e = max(s, t0), min(e, t1)
    if e > s:
        pts += [(s, 1, k), (e, -1, k)]
pts.sort()
cnt, acc, last = collections.Counter(), collections.Counter(), t0
for t, d, k in pts:
    acc[(min(cnt['conv'], 3), min(cnt['lstm'], 4))] += t - last
    last = t
    cnt[k] += d
tot = sum(acc.values())
print('window %.1f ms, %d recurrent launches' % (tot / 1e6, sum(1 for x in ls if t0 <= x < t1)))
for k in sorted(acc):
    if acc[k] / tot > 0.005:
        print('  full-chip kernels running: %d%s   recurrent kernels running: %d   %5.1f %%' % (k[0], '+' if k[0] == 3 else ' ', k[1], 100 * acc[k] / tot))
byc = collections.Counter()
for k, v in acc.items():
    byc[k[0]] += v
print('  by full-chip kernel count:', {k: round(100 * v / tot, 1) for k, v in sorted(byc.items())})
