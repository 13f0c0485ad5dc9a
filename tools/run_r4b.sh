set -u
O=gpurun_out/r4b; rm -rf $O; mkdir -p $O
(timeout 200 python tools/x3_check.py 2>&1 | grep -v amdgpu.ids > $O/x3_check.txt); cat $O/x3_check.txt
(timeout 300 python tools/kernel_ab.py "KRK_CONV_X3P=0" "KRK_CONV_X3P=1" "KRK_CONV_X3P=1,KRK_GEMM_STAG=8000" "KRK_CONV_X3P=1,KRK_GEMM_STAG=16000" "KRK_CONV_X3P=1,KRK_GEMM_STAG=24000" 2>&1 | grep -v amdgpu.ids > $O/ab.txt); cat $O/ab.txt
(KRAKEN_AMD_LIB=$PWD/kraken_amd/libkraken_amd_ablate.so timeout 300 python tools/kernel_ab.py "KRK_X3_DBG=0" "KRK_X3_DBG=4" "KRK_X3_DBG=1" "KRK_X3_DBG=2" "KRK_X3_DBG=5" "KRK_X3_DBG=7" 2>&1 | grep -v amdgpu.ids > $O/ab_ablate.txt); cat $O/ab_ablate.txt
(timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1); tail -3 $O/pytest_gpu.txt
python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20.json 2>/dev/null
for f in $O/bench*.json; do echo $f $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"); done
