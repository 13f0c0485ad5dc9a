#!/bin/bash
mkdir -p gpurun_out/r4s
timeout 900 python tools/lstm_narrow_probe.py > gpurun_out/r4s/lstm_narrow_probe.txt 2>&1
cat gpurun_out/r4s/lstm_narrow_probe.txt
