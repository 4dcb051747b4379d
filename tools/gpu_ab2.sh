#!/bin/bash
# Developer tool (GPU box): like tools/gpu_ab.sh but without the parity tests -- timing and
# FETCH_SIZE of one entry for each value of an environment variable.
set -e -o pipefail
VAR=$1; VALS=$2; ENTRY=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab2_$ENTRY
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$R
cd "$R"
for v in $VALS; do
  if [ "$v" = "unset" ]; then unset $VAR; else export $VAR=$v; fi
  echo "=== $VAR=$v"
  python3 tools/bench_kernel.py BENCHMARK3 $ENTRY 7 | grep -v "state built"
  (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/f_$v" -o ab --output-format csv -- \
    python3 "$R/tools/bench_kernel.py" BENCHMARK3 calib_stream,$ENTRY 2 > "$OUT/f_$v.log" 2> "$OUT/f_$v.err")
  python3 - "$OUT/f_$v/ab_counter_collection.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if sum(v) / len(v) > 3000:
        print(f"   FETCH x2 = {2*1024*sum(v)/len(v)/1e9:8.4f} GB  {k}")
PY
done
