set -u
O=gpurun_out/r4j; rm -rf $O; mkdir -p $O
(timeout 200 python tools/x3_check.py 2>&1 | grep -v amdgpu.ids > $O/x3_check.txt); cat $O/x3_check.txt
(timeout 300 python tools/kernel_ab.py "KRK_TAPS_DMA=0" "KRK_TAPS_DMA=1" 2>&1 | grep -v amdgpu.ids > $O/ab.txt); cat $O/ab.txt
(timeout 300 python tools/kernel_ab.py "KRK_TAPS_DMA=0" "KRK_TAPS_DMA=1" --ragged --n=200 --w=1000 2>&1 | grep -v amdgpu.ids > $O/ab_ragged.txt); cat $O/ab_ragged.txt
(timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1); tail -3 $O/pytest_gpu.txt
for rep in 1 2; do for d in 0 1; do
KRK_TAPS_DMA=$d python bench.py --no-cpu-baseline > $O/bench_dma${d}_$rep.json 2>/dev/null
echo dma=$d rep=$rep $(tail -1 $O/bench_dma${d}_$rep.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done; done
(timeout 200 python tools/bench_b_probe.py 2>&1 | grep -v amdgpu.ids > $O/bench_b.txt); cat $O/bench_b.txt
