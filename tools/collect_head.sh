#!/bin/bash
# Evidence at HEAD after the round's big collection (tools/collect_round.sh): what changed since -- the API path, the dewarp / prep kernels,
# the recurrent routing, the VGSL breadth -- plus the headline numbers again on this build.  One gpurun call:
#   bash tools/collect_head.sh r04   ->  gpurun_out/head_<tag>/
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/head_$TAG
rm -rf $O
mkdir -p $O
cd $R
md5sum kraken_amd/libkraken_amd.so > $O/${TAG}_head_lib_md5.txt
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/${TAG}_head_pytest_gpu.txt); tail -1 $O/${TAG}_head_pytest_gpu.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/${TAG}_head_smoke.txt); tail -1 $O/${TAG}_head_smoke.txt
python bench.py > $O/${TAG}_head_bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_head_bench_steps20.json 2>/dev/null
python bench.py --steps 1000 --no-cpu-baseline > $O/${TAG}_head_bench_steps1000.json 2>/dev/null
python bench.py --mode api --no-cpu-baseline > $O/${TAG}_head_bench_api.json 2>/dev/null
python bench.py --mode config4 --no-cpu-baseline > $O/${TAG}_head_bench_config4.json 2>/dev/null
python bench.py --gpus 2 --share-device --no-cpu-baseline > $O/${TAG}_head_two_ranks_one_device.json 2> $O/two_ranks.err
(timeout 200 python tools/bench_b_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_head_bench_b.txt)
(timeout 60 python tools/h2d_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_head_h2d_probe.txt)
(timeout 200 python tools/lstm_narrow_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_head_lstm_narrow_probe.txt)
for m in RGB L; do timeout 120 python tools/api_host_profile.py --workers 6 --mode $m 2>&1 | tail -1; done > $O/${TAG}_head_api_host_only.txt
(timeout 400 python tools/fuzz_plans.py ${FUZZ:-240} --time-seed 2>&1 | grep -v amdgpu.ids | tail -20 > $O/${TAG}_head_fuzz.txt)
for f in $O/${TAG}_head_*bench*.json $O/${TAG}_head_two_ranks*.json; do echo $(basename $f) $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('steps'))" 2>&1 | tail -1); done
tail -3 $O/${TAG}_head_fuzz.txt; cat $O/${TAG}_head_h2d_probe.txt
