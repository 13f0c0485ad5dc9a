#!/bin/bash
# Samples shader clock / package power (rocm-smi) while bench.py runs: pipelined (4 batches in flight) vs one stream.
for slots in 4 1; do
  python bench.py --steps $((slots == 4 ? 6000 : 2500)) --warmup 8 --slots $slots --no-cpu-baseline > gpurun_out/clk_bench_$slots.log 2>&1 &
  BP=$!
  sleep 11
  echo "--- slots $slots"
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Package Power" | head -3 | tr '\n' ' '; echo; sleep 1; done
  wait $BP
  tail -1 gpurun_out/clk_bench_$slots.log | cut -c1-110
done
