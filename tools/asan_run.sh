#!/bin/bash
# Host AddressSanitizer run of the C ABI's C++ side ("-fsanitize=address host build of the C++ glue").
#   bash tools/asan_run.sh [pytest args ...]        default: the GPU parity tests of layers, plans and fuzzed specs
# Builds kraken_amd/libkraken_amd_asan.so (python -m kraken_amd.build --asan: host code instrumented, device code as always),
# preloads clang's ASan runtime into python and points the package at that library.  Leak checking is off (python itself leaks
# by design at exit); everything else aborts the run with ASan's report.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
python -m kraken_amd.build --asan > /dev/null || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
export LD_PRELOAD="$RT" KRAKEN_AMD_LIB="$R/kraken_amd/libkraken_amd_asan.so"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0:exitcode=99"
if [ $# -eq 0 ]; then
  set -- tests/test_gpu_parity.py -m gpu -x -q -k "layers_against or forms or reshape or groups or hidden_sizes or edge_shapes or config4_ragged"
fi
exec python -m pytest "$@"
