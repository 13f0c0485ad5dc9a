#!/bin/bash
mkdir -p gpurun_out/r4q
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d['cases'].items():
    print(k, v['api_lines_per_s'], v.get('api_median_warm_pass'), v['api_all_passes'], v['engine_resident_input_lines_per_s'], v.get('gc'))
PY
}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dewarp or uploaded_page or rpred_reads or narrow" 2>&1 | tail -3 > gpurun_out/r4q/tests.txt
cat gpurun_out/r4q/tests.txt
timeout 600 python bench.py --mode api > gpurun_out/r4q/bench_api.json 2> gpurun_out/r4q/bench_api.err
show gpurun_out/r4q/bench_api.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4q/prof -o api -- python $GRAFT_REPO_ROOT/bench.py --mode api > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r4q/prof -type f ! -name "*kernel_stats*" -delete
