#!/bin/bash
# Developer tool (run on the GPU box through gpurun): rocprofv3 kernel statistics of
# bench.py plus the two PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass)
# over single entries, with a calibration copy of known byte count in the same run.
# Usage: tools/gpu_profile.sh TAG [CONFIG]      (CONFIG: BENCHMARK3 (default), BENCHMARK3_MPDATA, ...)
set -e -o pipefail
TAG=${1:-rXX}
CONFIG=${2:-BENCHMARK3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
KERNELS=calib_stream,step3d_t,rhs3d_tile,pre_step3d,step3d_uv,uv3dmix2,t3dmix2,prsgrd,rho_eos,omega,set_massflux,set_depth,lmd_vmix,bulk_flux,wvelocity,diag,step2d
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o "$TAG" --output-format csv -- \
  python3 "$R/bench.py" --config $CONFIG --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o "$TAG" --output-format csv -- \
  python3 "$R/tools/bench_kernel.py" $CONFIG $KERNELS 3 > "$OUT/pmc_fetch.log" 2> "$OUT/pmc_fetch.err"
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o "$TAG" --output-format csv -- \
  python3 "$R/tools/bench_kernel.py" $CONFIG $KERNELS 3 > "$OUT/pmc_write.log" 2> "$OUT/pmc_write.err"
echo "write pass done"
ls -R "$OUT" | head -40
