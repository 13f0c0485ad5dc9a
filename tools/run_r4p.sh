#!/bin/bash
# round 4: nested / parallel groups, Addition, x-axis summarising LSTMs on the GPU
mkdir -p gpurun_out/r4p
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "groups or golden or parallel or summarising or split_bf16_kernels_take or ending_in_an_image" 2>&1 | tail -25 > gpurun_out/r4p/groups_tests.txt
cat gpurun_out/r4p/groups_tests.txt
