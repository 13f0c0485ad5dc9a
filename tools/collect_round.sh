#!/bin/bash
# Everything the round's profiles/ directory holds, from ONE build, in one gpurun call:
#   bash tools/collect_round.sh r02      ->  gpurun_out/final_<tag>/   (copy what should be judged into profiles/)
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_$TAG
mkdir -p $O
cd $R
git rev-parse HEAD > $O/head.txt 2>/dev/null || true
md5sum kraken_amd/libkraken_amd.so > $O/lib_md5.txt
(timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1); tail -1 $O/pytest_gpu.txt
bash tools/profile_round.sh $TAG bf16x3 > $O/profile_round.log 2>&1
cp gpurun_out/prof_$TAG/kernel_stats.csv $O/${TAG}_bf16x3_kernel_stats.csv
cp gpurun_out/prof_$TAG/pmc_summary.json $O/${TAG}_bf16x3_pmc_summary.json
cp gpurun_out/prof_$TAG/bench_default.json $O/${TAG}_bf16x3_bench_default.json
bash tools/lstm_pmc.sh $O/lstm_pmc > $O/${TAG}_lstm_ws_pmc.txt 2>&1
S=3 G=2 bash tools/trace_run.sh > $O/${TAG}_bf16x3_trace_overlap.txt 2>&1
cd $R
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bf16x3_bench_steps20.json 2>/dev/null
python bench.py --precision bf16 --no-cpu-baseline > $O/${TAG}_bf16_optin_bench_default.json 2>/dev/null
python bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bf16_optin_bench_steps20.json 2>/dev/null
python bench.py --precision f32 --no-cpu-baseline > $O/${TAG}_f32_bench_default.json 2>/dev/null
python bench.py --mode config4 --no-cpu-baseline > $O/${TAG}_bench_config4.json 2>/dev/null
python bench.py --force-dist --no-cpu-baseline > $O/${TAG}_bench_force_dist.json 2> $O/force_dist.err
python bench.py --mode api --no-cpu-baseline > $O/${TAG}_bench_api.json 2>/dev/null
python tools/api_host_profile.py > $O/${TAG}_api_host_only.txt 2>&1
(timeout 200 python tools/lstm_ws_probe.py --quick > $O/${TAG}_lstm_ws_probe.txt 2>&1)
for f in $O/${TAG}_*bench*.json; do echo $(basename $f) $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('steps'))" 2>&1 | tail -1); done
