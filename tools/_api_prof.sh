cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06api4
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rpred or dewarp or bbox or api or page or prep" 2>&1 | tail -3 > gpurun_out/r06api4/tests.txt
python tools/cold_start_probe.py --passes 6 --workers 6 > gpurun_out/r06api4/L_w6.txt 2>&1
python tools/cold_start_probe.py --passes 6 --workers 6 --mode RGB > gpurun_out/r06api4/RGB_w6.txt 2>&1
python tools/cold_start_probe.py --passes 6 --workers 1 > gpurun_out/r06api4/L_w1.txt 2>&1
python bench.py --mode api > gpurun_out/r06api4/bench_api.json 2> gpurun_out/r06api4/bench_api.err
cat gpurun_out/r06api4/tests.txt; grep "^pass" gpurun_out/r06api4/*.txt; tail -14 gpurun_out/r06api4/L_w6.txt;  tail -12 gpurun_out/r06api4/RGB_w6.txt
