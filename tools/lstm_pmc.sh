#!/bin/bash
# PMC picture of the recurrent kernel alone (dev tool, run through gpurun): where its wave cycles go.
# Usage: bash tools/lstm_pmc.sh <out dir>   (counters in their own pass, no tracing domains)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=${1:-$R/gpurun_out/lstm_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/lstm_only.py <<PY
import sys, torch
sys.path.insert(0, '$R')
import kraken_amd
from kraken_amd.specs import BENCH_A
torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=BENCH_A); m.nn.set_precision('bf16x3'); m.to('cuda')
x = torch.rand(256, 1, 48, 1200).cuda()
for _ in range(3):
    m.nn(x); torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- python /tmp/lstm_only.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/p2 -- python /tmp/lstm_only.py > $O/p2.log 2>&1
python - <<PY
import csv, glob, collections
for p in ('p1', 'p2'):
    f = glob.glob('$O/%s/**/*counter_collection.csv' % p, recursive=True)
    if not f: print(p, 'no csv'); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "lstm" in r["Kernel_Name"]:
            d[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in d.items():
        print(p, k)
        for c, vals in v.items():
            print('   %-28s %14.0f  (n=%d)' % (c, sum(vals) / len(vals), len(vals)))
PY
