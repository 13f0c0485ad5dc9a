#!/bin/bash
# final: full GPU suite + smoke + default bench at HEAD
mkdir -p gpurun_out/r4x
md5sum kraken_amd/libkraken_amd.so > gpurun_out/r4x/lib_md5.txt
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r4x/pytest_gpu.txt); cat gpurun_out/r4x/pytest_gpu.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 > gpurun_out/r4x/smoke.txt); cat gpurun_out/r4x/smoke.txt
python bench.py --no-cpu-baseline > gpurun_out/r4x/bench_default.json 2>/dev/null; tail -1 gpurun_out/r4x/bench_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4x/bench_steps20.json 2>/dev/null; tail -1 gpurun_out/r4x/bench_steps20.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps20', d['value'], d['ms_per_step'])"
python bench.py --mode api --no-cpu-baseline > gpurun_out/r4x/bench_api.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4x/bench_api.json').read().strip().splitlines()[-1])
for k,v in d['cases'].items(): print(k, v['api_lines_per_s'], v.get('api_median_warm_pass'), v['engine_resident_input_lines_per_s'])
PY
