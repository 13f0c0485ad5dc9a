#!/bin/bash
mkdir -p gpurun_out/r4z
md5sum kraken_amd/libkraken_amd.so > gpurun_out/r4z/lib_md5.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4z/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r4z/bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/r4z/prof -name "*kernel_stats*" | head -1) gpurun_out/r4z/kernel_stats.csv
find gpurun_out/r4z/prof -type f -delete
tail -1 gpurun_out/r4z/bench_under_rocprof.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', d['value'], 'roofline kernel', r.get('kernel'), 'hip-event ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), r.get('frac_hip_events'), r.get('frac_rocprof'))"
head -6 gpurun_out/r4z/kernel_stats.csv | cut -c1-150
