#!/bin/bash
mkdir -p gpurun_out/r4u
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tap_kernel" 2>&1 | tail -3 > gpurun_out/r4u/tests.txt
cat gpurun_out/r4u/tests.txt
(timeout 200 python tools/fuzz_plans.py 75 --time-seed --only=5 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r4u/fuzz_two_gn.txt); cat gpurun_out/r4u/fuzz_two_gn.txt
(KRK_NO_C1GN=1 KRK_NO_CONV_X6=1 KRK_NO_TOSEQ_SPLIT=1 timeout 200 python tools/fuzz_plans.py 75 --time-seed --only=5 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r4u/fuzz_two_gn_round3_kernels.txt); cat gpurun_out/r4u/fuzz_two_gn_round3_kernels.txt
