set -u
O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
./tools/ubench/bufdma_probe > $O/bufdma_probe.txt 2>&1
(timeout 300 python tools/kernel_ab.py "KRK_GEMM_W=0" "KRK_GEMM_W=1" 2>&1 | grep -v amdgpu.ids > $O/ab_gemm.txt)
(timeout 120 python tools/kernel_ab.py "KRK_GEMM_W=0" "KRK_GEMM_W=1" --ragged --n=200 --w=1000 2>&1 | grep -v amdgpu.ids > $O/ab_gemm_ragged.txt)
(timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1); tail -3 $O/pytest_gpu.txt
KRK_GEMM_W=0 python bench.py --no-cpu-baseline > $O/bench_w0.json 2>/dev/null
python bench.py --no-cpu-baseline > $O/bench_w1.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_w1_s20.json 2>/dev/null
cat $O/bufdma_probe.txt $O/ab_gemm.txt $O/ab_gemm_ragged.txt
for f in $O/bench*.json; do echo $f $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); done
