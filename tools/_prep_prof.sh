cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06prep
for sh in 48 72; do
  python tools/prep_probe.py --src-h $sh > gpurun_out/r06prep/probe_$sh.txt 2>&1
  rocprofv3 --kernel-trace --stats -d gpurun_out/r06prep/p$sh -o p -- python tools/prep_probe.py --src-h $sh > gpurun_out/r06prep/prof_$sh.log 2>&1
done
python tools/prep_probe.py --src-h 90 --model-h 120 > gpurun_out/r06prep/probe_90_120.txt 2>&1
cat gpurun_out/r06prep/probe_*.txt
