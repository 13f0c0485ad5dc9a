#!/bin/bash
# The round-5 evidence at the FINAL library, ONE gpurun call:  bash tools/collect_r05.sh   ->  gpurun_out/final_r05/
# (copy what should be judged into profiles/).  Leaner than tools/collect_round.sh: the kernels of BENCH-A did not change in round 5,
# so the A/B matrices of round 4 are not repeated; new are the clock samples next to the PMC passes, the line-by-line parity run,
# the dense / strokes input sets, config 4 over 20 jobs, the preflight and the fuzz over the round-5 VGSL forms.
set -u
TAG=r05
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_$TAG
rm -rf $O
mkdir -p $O
cd $R
md5sum kraken_amd/libkraken_amd.so > $O/${TAG}_lib_md5.txt
(timeout 500 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu_full.txt 2>&1); tail -4 $O/${TAG}_pytest_gpu_full.txt > $O/${TAG}_pytest_gpu.txt; tail -1 $O/${TAG}_pytest_gpu.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/${TAG}_smoke.txt); tail -1 $O/${TAG}_smoke.txt
bash tools/profile_round.sh $TAG bf16x3 3 > $O/profile_round.log 2>&1
cp gpurun_out/prof_$TAG/kernel_stats.csv $O/${TAG}_bf16x3_kernel_stats.csv
cp gpurun_out/prof_$TAG/pmc_summary_load.json $O/${TAG}_bf16x3_pmc_summary_load.json
cp gpurun_out/prof_$TAG/pmc_summary_solo.json $O/${TAG}_bf16x3_pmc_summary_solo.json
cp gpurun_out/prof_$TAG/bench_default.json $O/${TAG}_bf16x3_bench_default.json
cp gpurun_out/prof_$TAG/clock_load.json $O/${TAG}_clock_load.json
cp gpurun_out/prof_$TAG/clock_solo.json $O/${TAG}_clock_solo.json
tail -1 gpurun_out/prof_$TAG/bench_under_rocprof.log > $O/${TAG}_bench_under_rocprof.json
cd $R
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bf16x3_bench_steps20.json 2>/dev/null
python bench.py --steps 4000 --no-cpu-baseline > $O/${TAG}_bench_steps4000.json 2>/dev/null
python bench.py --data dense --no-cpu-baseline > $O/${TAG}_bench_dense.json 2>/dev/null
python bench.py --data strokes --no-cpu-baseline > $O/${TAG}_bench_strokes.json 2>/dev/null
python bench.py --precision f32 --no-cpu-baseline > $O/${TAG}_f32_bench_default.json 2>/dev/null
python bench.py --precision bf16 --no-cpu-baseline > $O/${TAG}_bf16_optin_bench_default.json 2>/dev/null
python bench.py --mode config4 --no-cpu-baseline > $O/${TAG}_bench_config4.json 2>/dev/null
python bench.py --force-dist --no-cpu-baseline > $O/${TAG}_bench_force_dist.json 2> $O/force_dist.err
python bench.py --mode api --no-cpu-baseline > $O/${TAG}_bench_api.json 2>/dev/null
python bench.py --gpus 2 --share-device --no-cpu-baseline > $O/${TAG}_two_ranks_one_device.json 2> $O/two_ranks.err
PREFLIGHT_OUT=$O/preflight bash tools/scale_preflight.sh 20 > $O/${TAG}_preflight.log 2>&1; cp $O/preflight/summary.txt $O/${TAG}_preflight_summary.txt; cat $O/preflight/devices.txt >> $O/${TAG}_preflight_summary.txt
(timeout 200 python tools/lstm_ws_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_lstm_probe.txt)
(timeout 200 python tools/bench_b_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bench_b.txt)
(timeout 120 python tools/blla_forward.py --x3 2>&1 | grep -v amdgpu.ids | tail -30 > $O/${TAG}_blla.txt)
for i in 3 4 5; do (KRK_LSTM_V=3 timeout 200 python tools/ws_flake.py 1500 $i --slots 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_exchange_timeouts_three_in_flight.txt); done
(timeout 60 python tools/batch_invariance.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_batch_invariance.txt)
(timeout 260 python tools/fuzz_plans.py ${FUZZ:-200} --time-seed 2>&1 | grep -v amdgpu.ids | tail -30 > $O/${TAG}_fuzz.txt)
for f in $O/${TAG}_*bench*.json $O/${TAG}_two_ranks*.json; do echo $(basename $f) $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('steps'), d.get('parity_checked'))" 2>&1 | tail -1); done
tail -3 $O/${TAG}_fuzz.txt; cat $O/${TAG}_preflight_summary.txt | head -8; cat $O/${TAG}_exchange_timeouts_three_in_flight.txt
python -c "
import json; d=json.load(open('$O/${TAG}_bf16x3_pmc_summary_load.json')); print(json.dumps(d.get('clock'))); k=[(n,v) for n,v in d['kernels'].items() if 'mfma_util_chip' in v][:6]; [print(n, v.get('avg_us'), v.get('mfma_util_chip'), v.get('mfma_util_of_nominal_peak'), v.get('effective_clock_mhz'), v.get('hbm_read_MB_x2', v.get('hbm_read_MB_per_launch')), v.get('hbm_write_MB_per_launch')) for n,v in k]"
