set -u
O=gpurun_out/r4m; rm -rf $O; mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1); tail -6 $O/pytest_gpu.txt
(timeout 200 python tools/bench_b_probe.py 2>&1 | grep -v amdgpu.ids > $O/bench_b.txt); cat $O/bench_b.txt
(timeout 300 python tools/fuzz_plans.py 150 --time-seed 2>&1 | grep -v amdgpu.ids > $O/fuzz.txt); tail -15 $O/fuzz.txt
