set -u
O=gpurun_out/r4f; rm -rf $O; mkdir -p $O
(timeout 300 python -m pytest tests -m gpu -q -x -k "retried or dewarped_on_the_device or unsupported or golden" > $O/pytest_sel.txt 2>&1); tail -4 $O/pytest_sel.txt
KRK_PROFILE_API=1 python bench.py --mode api --no-cpu-baseline > $O/bench_api.json 2> $O/api_profile.txt
tail -c 1200 $O/bench_api.json; echo; grep -v amdgpu.ids $O/api_profile.txt | head -70
