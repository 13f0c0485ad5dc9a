#!/bin/bash
mkdir -p gpurun_out/r4z
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "rnd" 2>&1 | tail -12 > gpurun_out/r4z/tests_rnd.txt
cat gpurun_out/r4z/tests_rnd.txt
