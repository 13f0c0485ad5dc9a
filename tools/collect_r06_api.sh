#!/bin/bash
# The API-path part of the round-6 evidence again (Python-only changes after tools/collect_r06.sh):  bash tools/collect_r06_api.sh -> gpurun_out/final_r06_api/
set -u
TAG=r06
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_${TAG}_api
rm -rf $O; mkdir -p $O; cd $R
md5sum kraken_amd/libkraken_amd.so > $O/${TAG}_lib_md5.txt
(timeout 900 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu_full.txt 2>&1); tail -4 $O/${TAG}_pytest_gpu_full.txt > $O/${TAG}_pytest_gpu.txt; tail -1 $O/${TAG}_pytest_gpu.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/${TAG}_smoke.txt); tail -1 $O/${TAG}_smoke.txt
python bench.py --mode api --no-cpu-baseline > $O/${TAG}_bench_api.json 2>/dev/null
for n in 40 2048; do for md in L RGB; do (timeout 120 python tools/cold_start_probe.py --lines $n --mode $md --passes 6 2>&1 | grep -v amdgpu.ids > $O/${TAG}_cold_${md}_$n.txt); done; done
(timeout 120 python tools/cold_start_probe.py --lines 2048 --mode L --passes 6 --height 120 2>&1 | grep -v amdgpu.ids > $O/${TAG}_cold_L_2048_h120.txt)
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bf16x3_bench_steps20_again.json 2>/dev/null
tail -1 $O/${TAG}_bench_api.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); [print(k, c['api_lines_per_s'], c['api_best_pass'], c['api_all_passes']) for k,c in d['cases'].items()]"
grep "^pass" $O/${TAG}_cold_*.txt
tail -1 $O/${TAG}_bf16x3_bench_steps20_again.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
