#!/bin/bash
# full GPU suite at HEAD
mkdir -p gpurun_out/r4r
md5sum kraken_amd/libkraken_amd.so > gpurun_out/r4r/lib_md5.txt
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r4r/pytest_gpu.txt
cat gpurun_out/r4r/pytest_gpu.txt
