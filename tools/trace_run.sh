cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KRK_LSTM_G=${G:-1} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr1 -- python $R/bench.py --no-cpu-baseline --slots ${S:-4} > $R/gpurun_out/tr1.log 2>&1
python $R/tools/trace_overlap.py $(find $R/gpurun_out/tr1 -name "*kernel_trace.csv" | head -1)
tail -1 $R/gpurun_out/tr1.log | cut -c1-120
python - $(find $R/gpurun_out/tr1 -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:34]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print('%-36s n %4d avg_us %8.1f' % (k, len(v), sum(v) / len(v) / 1e3))
PY
