#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the DEFAULT bench command
#   2. three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GRBM_GUI_ACTIVE), short single-slot runs
# and condenses them with tools/summarize_pmc.py.  Usage: bash tools/profile_round.sh <tag> [precision]
set -u
TAG=${1:-r01}; PREC=${2:-bf16x3}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --precision $PREC --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $d -- python $R/bench.py --precision $PREC --steps 3 --warmup 1 --slots 1 --no-cpu-baseline > $d.log 2>&1
done
f() { dirname $(find $O/$1 -name "*$2" | head -1); }
python $R/tools/summarize_pmc.py $TAG $(f stats kernel_stats.csv) $(f pmc_FETCH_SIZE counter_collection.csv) $(f pmc_WRITE_SIZE counter_collection.csv) $(f pmc_SQ_VALU_MFMA_BUSY_CYCLES counter_collection.csv) > $O/pmc_summary.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python $R/bench.py --precision $PREC > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_under_rocprof.log | cut -c1-200
head -c 1500 $O/pmc_summary.json
