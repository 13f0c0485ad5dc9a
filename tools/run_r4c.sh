set -u
O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
(timeout 200 python tools/phase_stats.py 2>&1 | grep -v amdgpu.ids > $O/phase_stats.txt); cat $O/phase_stats.txt
(timeout 200 python tools/phase_stats.py KRK_GEMM_NBUF=4 2>&1 | grep -v amdgpu.ids > $O/phase_stats_nbuf4.txt); cat $O/phase_stats_nbuf4.txt
(timeout 300 python tools/kernel_ab.py "KRK_GEMM_NBUF=3" "KRK_GEMM_NBUF=4" "KRK_GEMM_W=0" 2>&1 | grep -v amdgpu.ids > $O/ab.txt); cat $O/ab.txt
(KRAKEN_AMD_LIB=$PWD/kraken_amd/libkraken_amd_ablate.so timeout 300 python tools/kernel_ab.py "KRK_X3_DBG=0" "KRK_X3_DBG=6" "KRK_X3_DBG=8" "KRK_X3_DBG=14" 2>&1 | grep -v amdgpu.ids > $O/ab_ablate.txt); cat $O/ab_ablate.txt
(timeout 300 python -m pytest tests -m gpu -q -x -k "two_ranks or rccl" > $O/pytest_two_ranks.txt 2>&1); tail -5 $O/pytest_two_ranks.txt
python bench.py --gpus 2 --share-device --no-cpu-baseline > $O/two_ranks_one_device.json 2> $O/two_ranks.err; tail -c 1500 $O/two_ranks_one_device.json; tail -3 $O/two_ranks.err
