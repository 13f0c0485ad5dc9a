set -u
O=gpurun_out/r4o; rm -rf $O; mkdir -p $O
python bench.py --steps 4000 --warmup 8 --no-cpu-baseline > $O/r04_bench_steps4000.json 2>/dev/null
tail -1 $O/r04_bench_steps4000.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('4000 steps', d['value'], d['ms_per_step'], d['gathered_lines'])"
(timeout 400 python tools/fuzz_plans.py 300 --time-seed 2>&1 | grep -v amdgpu.ids > $O/r04_fuzz_300s.txt); head -3 $O/r04_fuzz_300s.txt
python bench.py > $O/r04_bench_default_with_cpu_baseline.json 2>/dev/null
tail -1 $O/r04_bench_default_with_cpu_baseline.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['step_traffic'], d['roofline'].get('mfma_busy_on_its_CUs'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
