cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 1 4; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s$s -- python $R/bench.py --steps 12 --warmup 4 --slots $s --no-cpu-baseline > $R/gpurun_out/prof_s$s.log 2>&1
f=$(find $R/gpurun_out/prof_s$s -name "*kernel_stats.csv" | head -1); echo "== slots $s"; cut -d, -f1-4 $f | cut -c1-110 | head -14; tail -1 $R/gpurun_out/prof_s$s.log | cut -c1-100
done
