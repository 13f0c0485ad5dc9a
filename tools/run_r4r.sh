#!/bin/bash
# full GPU suite at HEAD
mkdir -p gpurun_out/r4r
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r4r/pytest_gpu.txt
cat gpurun_out/r4r/pytest_gpu.txt
