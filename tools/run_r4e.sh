set -u
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for w in 0 1; do for p in 0 1; do
KRK_GEMM_W=$w KRK_CONV_X3P=$p python bench.py --no-cpu-baseline > $O/bench_w${w}_p${p}_$rep.json 2>/dev/null
echo w=$w p=$p rep=$rep $(tail -1 $O/bench_w${w}_p${p}_$rep.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done; done
