#!/bin/bash
# PMC traffic of a few entries only: tools/gpu_pmc_one.sh TAG KERNELS [name=value ...] (comma list and parameter
# settings for tools/bench_kernel.py, e.g. uv3dmix2 uv_vis2=2)
set -e -o pipefail
TAG=$1
KERNELS=calib_stream,$2
shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o "$TAG" --output-format csv -- \
  python3 "$R/tools/bench_kernel.py" BENCHMARK3 $KERNELS 3 "$@" > "$OUT/pmc_fetch.log" 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o "$TAG" --output-format csv -- \
  python3 "$R/tools/bench_kernel.py" BENCHMARK3 $KERNELS 3 "$@" > "$OUT/pmc_write.log" 2> "$OUT/pmc_write.err"
echo done
