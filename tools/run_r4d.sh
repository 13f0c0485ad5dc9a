set -u
O=gpurun_out/r4d; rm -rf $O; mkdir -p $O
(timeout 200 python tools/x3_check.py 2>&1 | grep -v amdgpu.ids > $O/x3_check.txt); cat $O/x3_check.txt
(timeout 200 python tools/phase_stats.py 2>&1 | grep -v amdgpu.ids > $O/phase_stats.txt); cat $O/phase_stats.txt
(timeout 300 python tools/kernel_ab.py "KRK_GEMM_W=1" "KRK_GEMM_W=0,KRK_CONV_X3P=0" 2>&1 | grep -v amdgpu.ids > $O/ab.txt); cat $O/ab.txt
for i in 0 1 2; do (KRK_LSTM_V=3 timeout 200 python tools/ws_flake.py 150 $i 2>&1 | grep -v amdgpu.ids >> $O/ws_forced_narrow.txt); done; cat $O/ws_forced_narrow.txt
(timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1); tail -3 $O/pytest_gpu.txt
python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20.json 2>/dev/null
KRK_GEMM_W=0 KRK_CONV_X3P=0 python bench.py --no-cpu-baseline > $O/bench_old_kernels.json 2>/dev/null
for f in $O/bench*.json; do echo $f $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); done
