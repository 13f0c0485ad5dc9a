#!/bin/bash
mkdir -p gpurun_out/r4v
python bench.py --mode api --no-cpu-baseline > gpurun_out/r4v/bench_api.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4v/bench_api.json').read().strip().splitlines()[-1])
for k,v in d['cases'].items(): print(k, v['api_lines_per_s'], v.get('api_median_warm_pass'), v['api_all_passes'], v['engine_resident_input_lines_per_s'])
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rpred or mm_rpred or predict or dewarped_on_the_device" 2>&1 | tail -2
