#!/bin/bash
# Dev tool (GPU): per-kernel times of the line-preparation kernels (tools/prep_probe.py under rocprofv3 --kernel-trace), source heights 48 and 72.
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/prepk
for sh in 48 72; do
  rocprofv3 --kernel-trace -d gpurun_out/prepk/p$sh -o p -- python tools/prep_probe.py --src-h $sh > gpurun_out/prepk/prof_$sh.log 2>&1
done
python - <<'PY'
import sqlite3
for h in (48, 72):
    db = sqlite3.connect(f'gpurun_out/prepk/p{h}/p_results.db')
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%dw_%' or s.kernel_name like '%prep_lines%' group by s.kernel_name order by 3 desc").fetchall()
    print('source height', h)
    tot = 0
    for n, c, a in rows:
        short = n.split('GLOBAL__N_1')[1][2:24] if 'GLOBAL__N_1' in n else n[:22]
        print(f'   {short:24s} calls {c:3d} avg {a/1e3:8.1f} us')
        if 'dw_' in n: tot += a
    print('   dewarp kernels total %.1f us per 256 lines' % (tot / 1e3))
PY
