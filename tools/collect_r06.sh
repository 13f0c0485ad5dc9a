#!/bin/bash
# The round-6 evidence at the FINAL library, ONE gpurun call:  bash tools/collect_r06.sh   ->  gpurun_out/final_r06/
# (copy what should be judged into profiles/).  bench.py now profiles itself (rocprofv3 stats + three --pmc passes after the timed
# region), so the headline line carries this run's counters; tools/profile_round.sh still adds the solo / load split and the clock.
set -u
TAG=r06
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_$TAG
rm -rf $O
mkdir -p $O
cd $R
md5sum kraken_amd/libkraken_amd.so > $O/${TAG}_lib_md5.txt
(timeout 600 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu_full.txt 2>&1); tail -4 $O/${TAG}_pytest_gpu_full.txt > $O/${TAG}_pytest_gpu.txt; tail -1 $O/${TAG}_pytest_gpu.txt
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "every_line_against_kraken or distinct_lines_against_kraken or height_120" 2>&1 | grep -E "^\[(f32|bf16x3)\]|passed|failed" > $O/${TAG}_pytest_line_by_line.txt); cat $O/${TAG}_pytest_line_by_line.txt
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/${TAG}_smoke.txt); tail -1 $O/${TAG}_smoke.txt
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bf16x3_bench_steps20.json 2> $O/bench20.err
python bench.py > $O/${TAG}_bf16x3_bench_default.json 2>/dev/null
python bench.py --steps 4000 --no-cpu-baseline --no-self-profile > $O/${TAG}_bench_steps4000.json 2>/dev/null
KRK_X3P_SB=0 python bench.py --no-cpu-baseline --no-self-profile --no-config4-check > $O/${TAG}_bench_default_two_tile_buffers.json 2>/dev/null
python bench.py --no-cpu-baseline --no-self-profile --no-config4-check > $O/${TAG}_bench_default_again.json 2>/dev/null
bash tools/profile_round.sh $TAG bf16x3 3 > $O/profile_round.log 2>&1
cp gpurun_out/prof_$TAG/kernel_stats.csv $O/${TAG}_bf16x3_kernel_stats.csv
cp gpurun_out/prof_$TAG/pmc_summary_load.json $O/${TAG}_bf16x3_pmc_summary_load.json
cp gpurun_out/prof_$TAG/pmc_summary_solo.json $O/${TAG}_bf16x3_pmc_summary_solo.json
cp gpurun_out/prof_$TAG/clock_load.json $O/${TAG}_clock_load.json
cp gpurun_out/prof_$TAG/clock_solo.json $O/${TAG}_clock_solo.json
cd $R
python bench.py --precision f32 --no-cpu-baseline --no-self-profile > $O/${TAG}_f32_bench_default.json 2>/dev/null
python bench.py --precision bf16 --no-cpu-baseline --no-self-profile > $O/${TAG}_bf16_optin_bench_default.json 2>/dev/null
python bench.py --mode config4 --no-cpu-baseline > $O/${TAG}_bench_config4.json 2>/dev/null
python bench.py --force-dist --no-cpu-baseline --no-self-profile > $O/${TAG}_bench_force_dist.json 2> $O/force_dist.err
python bench.py --mode api --no-cpu-baseline > $O/${TAG}_bench_api.json 2>/dev/null
python bench.py --gpus 2 --share-device --no-cpu-baseline --no-self-profile > $O/${TAG}_two_ranks_one_device.json 2> $O/two_ranks.err
PREFLIGHT_OUT=$O/preflight bash tools/scale_preflight.sh 20 > $O/${TAG}_preflight.log 2>&1; cp $O/preflight/summary.txt $O/${TAG}_preflight_summary.txt; cat $O/preflight/devices.txt >> $O/${TAG}_preflight_summary.txt
(timeout 200 python tools/kernel_ab.py "KRK_X3P_SB=0" "KRK_X3P_SB=1" --rounds=9 2>&1 | grep -v amdgpu.ids > $O/${TAG}_kernel_ab_tile_buffers.txt)
(timeout 200 python tools/height120_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_height120.txt)
(timeout 200 python tools/bench_b_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bench_b.txt)
(timeout 120 python tools/blla_forward.py --x3 2>&1 | grep -v amdgpu.ids | tail -30 > $O/${TAG}_blla.txt)
for i in 3 4; do (KRK_LSTM_V=3 timeout 200 python tools/ws_flake.py 1500 $i --slots 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_exchange_timeouts_three_in_flight.txt); done
(timeout 260 python tools/fuzz_plans.py ${FUZZ:-150} --time-seed 2>&1 | grep -v amdgpu.ids | tail -30 > $O/${TAG}_fuzz.txt)
for n in 40 2048; do for md in L RGB; do (timeout 120 python tools/cold_start_probe.py --lines $n --mode $md --passes 6 2>&1 | grep -v amdgpu.ids > $O/${TAG}_cold_${md}_$n.txt); done; done
(timeout 200 bash tools/prep_kernels.sh 2>&1 | grep -v amdgpu.ids > $O/${TAG}_prep_kernels_final.txt)
for f in $O/${TAG}_*bench*.json $O/${TAG}_two_ranks*.json; do echo $(basename $f) $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('steps'), d.get('parity_checked'))" 2>&1 | tail -1 | cut -c1-400); done
tail -3 $O/${TAG}_fuzz.txt; head -8 $O/${TAG}_preflight_summary.txt; cat $O/${TAG}_exchange_timeouts_three_in_flight.txt; cat $O/${TAG}_height120.txt | cut -c1-400; grep "pass 0" $O/${TAG}_cold_*.txt
