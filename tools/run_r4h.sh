set -u
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
for v in 0 1; do
[ $v = 1 ] && export KRK_API_NOGC=1
python bench.py --mode api --no-cpu-baseline > $O/bench_api_nogc$v.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/bench_api_nogc$v.json').read().strip().splitlines()[-1])
for k,v in d['cases'].items(): print('nogc=$v', k, v['api_lines_per_s'], v['api_all_passes'])"
done
