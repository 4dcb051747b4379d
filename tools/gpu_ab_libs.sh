#!/bin/bash
# Developer tool (GPU box): A/B several builds of the library (python -m roms_trunk_mgh_amd._build <variant> -D...)
# on one entry: parity tests of the entry, then hipEvent timing on a configuration.
# Usage: tools/gpu_ab_libs.sh ENTRY TEST_FILTER CONFIG variant1 variant2 ...     ("base" = the product library)
set -o pipefail
ENTRY=$1; FILT=$2; CONFIG=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
export PYTHONPATH=$R
for v in "$@"; do
  if [ "$v" = base ]; then LIBARG=""; else LIBARG="--lib $R/roms_trunk_mgh_amd/libroms_hip_$v.so"; fi
  echo "=== $v"
  # the parity tests always run on the product library (no environment variable can redirect it)
  [ "$v" = base ] && python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_main3d.py -m gpu -x -q -k "$FILT" 2>&1 | tail -1
  python3 tools/bench_kernel.py $LIBARG $CONFIG $ENTRY 9 | grep -v "state built"
done
