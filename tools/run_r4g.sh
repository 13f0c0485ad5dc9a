set -u
O=gpurun_out/r4g; rm -rf $O; mkdir -p $O
(timeout 400 python -m pytest tests -m gpu -q -x -k "dewarp or rpred or predict or retried or mm_rpred or records" > $O/pytest_sel.txt 2>&1); tail -4 $O/pytest_sel.txt
KRK_PROFILE_API=1 python bench.py --mode api --no-cpu-baseline > $O/bench_api.json 2> $O/api_profile.txt
python -c "
import json; d=json.loads(open('$O/bench_api.json').read().strip().splitlines()[-1])
for k,v in d['cases'].items(): print(k, v['api_lines_per_s'], v['api_all_passes'], v['engine_resident_input_lines_per_s'])"
grep -v amdgpu.ids $O/api_profile.txt | head -24
