#!/bin/bash
mkdir -p gpurun_out/r4t
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tap_kernel" 2>&1 | tail -6 > gpurun_out/r4t/tests.txt
cat gpurun_out/r4t/tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4t/prof -o api -- python $GRAFT_REPO_ROOT/bench.py --mode api > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r4t/prof -type f ! -name "*kernel_stats*" -delete
