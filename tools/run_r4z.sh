#!/bin/bash
mkdir -p gpurun_out/r4z
(timeout 260 python tools/fuzz_plans.py 170 --time-seed 2>&1 | grep -v amdgpu.ids | tail -20 > gpurun_out/r4z/fuzz.txt); cat gpurun_out/r4z/fuzz.txt
