cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06api2
python tools/cold_start_probe.py --passes 6 --workers 6 > gpurun_out/r06api2/L_w6.txt 2>&1
python tools/cold_start_probe.py --passes 6 --workers 6 --mode RGB > gpurun_out/r06api2/RGB_w6.txt 2>&1
python bench.py --mode api > gpurun_out/r06api2/bench_api.json 2> gpurun_out/r06api2/bench_api.err
rocprofv3 --kernel-trace -d gpurun_out/r06api2/prof -o api -- python tools/cold_start_probe.py --passes 6 --workers 6 > gpurun_out/r06api2/prof.log 2>&1
grep "^pass" gpurun_out/r06api2/*.txt gpurun_out/r06api2/prof.log
