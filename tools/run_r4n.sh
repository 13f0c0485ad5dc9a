set -u
O=gpurun_out/r4n; rm -rf $O; mkdir -p $O
(timeout 300 python tools/kernel_ab.py "KRK_GEMM_SPREAD=0" "KRK_GEMM_SPREAD=1" 2>&1 | grep -v amdgpu.ids > $O/ab.txt); cat $O/ab.txt
(timeout 300 python tools/kernel_ab.py "KRK_GEMM_SPREAD=0" "KRK_GEMM_SPREAD=1" --ragged --n=200 --w=1000 2>&1 | grep -v amdgpu.ids > $O/ab_ragged.txt); cat $O/ab_ragged.txt
for rep in 1 2; do for d in 0 1; do
echo spread=$d rep=$rep $(KRK_GEMM_SPREAD=$d python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done
(timeout 400 python -m pytest tests -m gpu -q -x -k "x3 or golden or bench" > $O/pytest_sel.txt 2>&1); tail -3 $O/pytest_sel.txt
