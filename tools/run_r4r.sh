#!/bin/bash
# full GPU suite at HEAD + a short fuzz run with the group specs
mkdir -p gpurun_out/r4r
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r4r/pytest_gpu.txt
cat gpurun_out/r4r/pytest_gpu.txt
timeout 200 python tools/fuzz_plans.py 90 --time-seed 2>&1 | tail -18 > gpurun_out/r4r/fuzz_90s.txt
cat gpurun_out/r4r/fuzz_90s.txt
