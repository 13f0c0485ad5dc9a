#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the DEFAULT bench command (the benchmark's slot count)
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + GRBM_GUI_ACTIVE), each TWICE: "solo" (--slots 1: one batch
#      in flight, the kernel has the chip to itself) and "load" (the benchmark's slot count: what the durations of 1. saw)
# and condenses them with tools/summarize_pmc.py.  Output directories are emptied first and files are picked by newest
# mtime, so a summary can only come from THIS run.   Usage: bash tools/profile_round.sh <tag> [precision] [slots]
set -u
TAG=${1:-r03}; PREC=${2:-bf16x3}; SLOTS=${3:-3}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$TAG
rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
newest() { find $O/$1 -name "*$2" -printf '%T@ %p\n' | sort -n | tail -1 | cut -d' ' -f2-; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --precision $PREC --slots $SLOTS --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
for mode in solo load; do
  s=$([ $mode = solo ] && echo 1 || echo $SLOTS)
  # the clock and package power of an UN-profiled run of the same command, long enough to sample
  python $R/tools/clock_sample.py $O/clock_$mode.json -- python $R/bench.py --precision $PREC --steps 2500 --warmup 8 --slots $s --no-cpu-baseline > $O/clock_$mode.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    d=$O/pmc_${mode}_$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --output-format csv -d $d -- python $R/bench.py --precision $PREC --steps 6 --warmup 2 --slots $s --no-cpu-baseline > $d.log 2>&1
  done
  python $R/tools/summarize_pmc.py $TAG-$mode "$(newest stats kernel_stats.csv)" "$(newest pmc_${mode}_FETCH_SIZE counter_collection.csv)" \
      "$(newest pmc_${mode}_WRITE_SIZE counter_collection.csv)" "$(newest pmc_${mode}_SQ_VALU_MFMA_BUSY_CYCLES counter_collection.csv)" $O/clock_$mode.json > $O/pmc_summary_$mode.json
done
cp "$(newest stats kernel_stats.csv)" $O/kernel_stats.csv
python $R/bench.py --precision $PREC --slots $SLOTS > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_under_rocprof.log | cut -c1-200
head -c 1200 $O/pmc_summary_load.json
