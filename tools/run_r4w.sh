#!/bin/bash
mkdir -p gpurun_out/r4w
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "image_lstm or blla or segment or groups" 2>&1 | tail -2 > gpurun_out/r4w/tests.txt
cat gpurun_out/r4w/tests.txt
(timeout 300 python tools/blla_forward.py --x3 --layers 2>&1 | grep -v amdgpu.ids > gpurun_out/r4w/blla_x3.txt); head -2 gpurun_out/r4w/blla_x3.txt; grep lstm_rec gpurun_out/r4w/blla_x3.txt
