#!/bin/bash
# Everything the round's profiles/ directory holds, from ONE build, in one gpurun call:
#   bash tools/collect_round.sh r04      ->  gpurun_out/final_<tag>/   (copy what should be judged into profiles/)
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final_$TAG
rm -rf $O
mkdir -p $O
cd $R
git rev-parse HEAD > $O/head.txt 2>/dev/null || true
md5sum kraken_amd/libkraken_amd.so > $O/lib_md5.txt
(timeout 600 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu.txt 2>&1); tail -1 $O/${TAG}_pytest_gpu.txt
bash tools/profile_round.sh $TAG bf16x3 3 > $O/profile_round.log 2>&1
cp gpurun_out/prof_$TAG/kernel_stats.csv $O/${TAG}_bf16x3_kernel_stats.csv
cp gpurun_out/prof_$TAG/pmc_summary_load.json $O/${TAG}_bf16x3_pmc_summary_load.json
cp gpurun_out/prof_$TAG/pmc_summary_solo.json $O/${TAG}_bf16x3_pmc_summary_solo.json
cp gpurun_out/prof_$TAG/bench_default.json $O/${TAG}_bf16x3_bench_default.json
bash tools/lstm_pmc.sh $O/lstm_pmc > $O/${TAG}_lstm_ws_pmc.txt 2>&1
cd $R
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bf16x3_bench_steps20.json 2>/dev/null
for sl in 2 4; do python bench.py --slots $sl --no-cpu-baseline > $O/${TAG}_bf16x3_bench_slots$sl.json 2>/dev/null; done
python bench.py --host-input --no-cpu-baseline > $O/${TAG}_bf16x3_bench_host_input.json 2>/dev/null
python bench.py --precision bf16 --no-cpu-baseline > $O/${TAG}_bf16_optin_bench_default.json 2>/dev/null
python bench.py --precision f32 --no-cpu-baseline > $O/${TAG}_f32_bench_default.json 2>/dev/null
python bench.py --mode config4 --no-cpu-baseline > $O/${TAG}_bench_config4.json 2>/dev/null
python bench.py --force-dist --no-cpu-baseline > $O/${TAG}_bench_force_dist.json 2> $O/force_dist.err
python bench.py --mode api --no-cpu-baseline > $O/${TAG}_bench_api.json 2>/dev/null
(timeout 200 python tools/bench_b_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bench_b.txt)
(KRK_NO_C1GN=1 KRK_NO_CONV_X6=1 KRK_NO_TOSEQ_SPLIT=1 timeout 200 python tools/bench_b_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bench_b_round3_kernels.txt)
python bench.py --gpus 2 --share-device --no-cpu-baseline > $O/${TAG}_two_ranks_one_device.json 2> $O/two_ranks.err
(timeout 300 python tools/kernel_ab.py "KRK_CONV_X3P=0,KRK_TAPS_DMA=0" "KRK_CONV_X3P=1,KRK_TAPS_DMA=1" 2>&1 | grep -v amdgpu.ids > $O/${TAG}_kernel_ab_alone.txt)
for rep in 1 2; do for cfg in "0 0 0" "0 1 0" "0 1 1" "1 1 1"; do set -- $cfg
  echo "KRK_CONV_X3P=$2 KRK_TAPS_DMA=$3 rep $rep:" $(KRK_CONV_X3P=$2 KRK_TAPS_DMA=$3 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'lines/s', d['ms_per_step'], 'ms/step')") >> $O/${TAG}_kernel_matrix.txt
done; done
[ -f kraken_amd/libkraken_amd_ablate.so ] || python -m kraken_amd.build --ablate > /dev/null 2>&1
(timeout 200 python tools/phase_stats.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_phase_stats.txt)
for i in 0 1 2; do (KRK_LSTM_V=3 timeout 300 python tools/ws_flake.py 350 $i 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_exchange_timeouts_lstm_ws_forced_narrow.txt); done
(KRK_CONV_X6=0 timeout 200 python tools/fuzz_plans.py 60 2>&1 | grep -v amdgpu.ids > $O/${TAG}_fuzz_f32_convs_same_seed.txt)
(timeout 200 python tools/fuzz_plans.py 60 2>&1 | grep -v amdgpu.ids > $O/${TAG}_fuzz_x6_convs_same_seed.txt)
(timeout 200 python tools/lstm_ws_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_lstm_probe.txt)
(timeout 150 python tools/ws_flake.py 100 2>&1 | grep -v amdgpu.ids > $O/${TAG}_exchange_timeouts_default.txt)
(timeout 60 python tools/batch_invariance.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_batch_invariance.txt)
(timeout 200 python tools/fuzz_plans.py ${FUZZ:-100} --time-seed 2>&1 | grep -v amdgpu.ids > $O/${TAG}_fuzz.txt)
for f in $O/${TAG}_*bench*.json; do echo $(basename $f) $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d.get('steps'))" 2>&1 | tail -1); done
