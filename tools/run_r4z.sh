#!/bin/bash
mkdir -p gpurun_out/r4z
python bench.py --steps 4000 --no-cpu-baseline > gpurun_out/r4z/bench_steps4000.json 2>/dev/null
tail -1 gpurun_out/r4z/bench_steps4000.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('4000 steps', d['value'], d['ms_per_step'])"
