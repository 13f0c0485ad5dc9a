#!/bin/bash
# First contact with a multi-GPU node: what is visible, and bench.py --gpus N for N = 1, 2, 4, ... <= visible devices (one process per
# GPU, RCCL over xGMI), each line checked: every rank in the collective, the collective on RCCL ("nccl"), every rank's own lines/s and
# gather time logged.  Writes gpurun_out/preflight/ (scale_N.json per N, summary.txt); exit code 0 only when every N passed.
#   bash tools/scale_preflight.sh [steps]            on a GPU box
#   PREFLIGHT_STUB=1 bash tools/scale_preflight.sh   host stub in place of the engine, gloo (the CPU test-suite runs this)
set -u
STEPS=${1:-20}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=${PREFLIGHT_OUT:-$R/gpurun_out/preflight}
mkdir -p "$O"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
STUB=${PREFLIGHT_STUB:-0}
if [ "$STUB" = 1 ]; then
    VISIBLE=${PREFLIGHT_RANKS:-2}
    EXTRA="--stub-engine --batch 8 --width 64"
    WANT_BACKEND=gloo
else
    VISIBLE=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null | tail -1)
    EXTRA="--no-cpu-baseline"
    WANT_BACKEND=nccl
    (rocm-smi --showtopo 2>/dev/null || true) > "$O/topology.txt"
    python - > "$O/devices.txt" 2>&1 <<'PY'
import torch
from kraken_amd import dist as kdist
n = torch.cuda.device_count()
ids = kdist._device_pci_ids(n)
for i, (pid, node) in enumerate(zip(ids, kdist.device_numa_nodes(ids))):
    p = torch.cuda.get_device_properties(i)
    print(f'device {i}: {p.name} {p.total_memory >> 30} GiB pci {pid} numa_node {node}')
PY
    cat "$O/devices.txt"
fi
echo "visible devices: $VISIBLE" | tee "$O/summary.txt"
[ "${VISIBLE:-0}" -ge 1 ] || { echo "no device visible" | tee -a "$O/summary.txt"; exit 1; }
FAIL=0
N=1
while [ "$N" -le "$VISIBLE" ]; do
    PORT=$((29600 + N))
    if [ "$N" = 1 ]; then
        python bench.py --gpus 1 --steps "$STEPS" --warmup 5 $EXTRA > "$O/scale_$N.json" 2> "$O/scale_$N.err"
    else
        python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
            bench.py --gpus "$N" --steps "$STEPS" --warmup 5 $EXTRA > "$O/scale_$N.json" 2> "$O/scale_$N.err"
    fi
    RC=$?
    python - "$O/scale_$N.json" "$N" "$WANT_BACKEND" "$RC" <<'PY' | tee -a "$O/summary.txt"
import json, sys
path, n, backend, rc = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
except Exception as e:
    print(f'N={n}: FAILED (rc {rc}): no JSON line ({type(e).__name__}); see {path[:-5]}.err')
    sys.exit(0)
bad = []
if rc:
    bad.append(f'rc {rc}')
if d.get('n_gpus') != n:
    bad.append(f"n_gpus {d.get('n_gpus')}")
if d.get('ranks_in_collective') != n:
    bad.append(f"ranks_in_collective {d.get('ranks_in_collective')}")
if n > 1 and d.get('collective_backend') != backend:
    bad.append(f"collective_backend {d.get('collective_backend')} (want {backend})")
if d.get('gathered_lines') != n * d['steps'] * d['config']['lines_per_gpu_step']:
    bad.append(f"gathered_lines {d.get('gathered_lines')}")
pr = d.get('per_rank') or {'lines_per_s': [d['value']], 'gather_ms': [d['gather_ms']]}
print(f"N={n}: {'FAILED: ' + ', '.join(bad) if bad else 'ok'}  {d['value']} lines/s whole job, {d['ms_per_step']} ms/step, "
      f"per rank {pr['lines_per_s']} lines/s, gather {pr['gather_ms']} ms, host us/line per rank {pr.get('host_us_per_line', [d.get('host_us_per_line', {}).get('codec_strings')])}, "
      f"cpus per rank {d.get('host_cpus_per_rank')}")
PY
    grep -q "^N=$N: ok" "$O/summary.txt" || FAIL=1
    N=$((N * 2))
done
[ "$FAIL" = 0 ] && echo "preflight passed" | tee -a "$O/summary.txt" || echo "preflight FAILED" | tee -a "$O/summary.txt"
exit $FAIL
