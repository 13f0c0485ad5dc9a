"""Dev tool: device-busy time and the batch timeline of the LAST rpred() pass in a rocprofv3 --kernel-trace database
(rocprofv3 --kernel-trace -d DIR -o api -- python tools/cold_start_probe.py ...):  python tools/api_timeline.py DIR/api_results.db [--rgb]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rgb = '--rgb' in sys.argv
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = db.execute(f"select d.start, d.end, s.kernel_name, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
t0 = rows[0][0]
ev = [(a - t0, b - t0, n, q) for a, b, n, q in rows]
mark = 'prep_lines' if rgb else 'dw_minmax'
mm = [e for e in ev if mark in e[2]]
lo = mm[-8][0] - 0.2e6
busy, cs, ce = 0, None, None
for a, b, n, q in ev:
    if a < lo:
        continue
    if ce is None or a > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
print('last pass: span %.2f ms, device busy %.2f ms' % ((ce - lo) / 1e6, busy / 1e6))
for a, b, n, q in ev:
    if a < lo:
        continue
    for key in ('dw_minmax', 'dw_spread', 'dw_apply', 'prep_lines', 'conv1_x3_kernelILi3ELb1ELb1ELb1ELi3', 'collapse', 'copyBuffer'):
        if key in n and (key != 'copyBuffer' or b - a > 0.2e6):
            print(f"{(a - lo) / 1e6:7.2f} -> {(b - lo) / 1e6:7.2f} q{q} {key[:10]}")
