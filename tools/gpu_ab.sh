#!/bin/bash
# Developer tool (GPU box): A/B a kernel's variants selected by an environment variable.
# For each value: parity tests of the entry, hipEvent timing on BENCHMARK3, FETCH_SIZE pass.
# Usage: tools/gpu_ab.sh ENVVAR "v0 v1 ..." entry test_filter
set -e -o pipefail
VAR=$1; VALS=$2; ENTRY=$3; FILT=$4
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_$ENTRY
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$R
cd "$R"
for v in $VALS; do
  export $VAR=$v
  echo "=== $VAR=$v"
  python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "$FILT" 2>&1 | tail -1
  python3 tools/bench_kernel.py BENCHMARK3 $ENTRY 7 | grep -v "state built"
  (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/f_$v" -o ab --output-format csv -- \
    python3 "$R/tools/bench_kernel.py" BENCHMARK3 calib_stream,$ENTRY 2 > "$OUT/f_$v.log" 2> "$OUT/f_$v.err")
  python3 - "$OUT/f_$v/ab_counter_collection.csv" <<'EOF'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if sum(v) / len(v) > 3000:
        print(f"   FETCH x2 = {2*1024*sum(v)/len(v)/1e9:8.4f} GB  {k}")
EOF
done
