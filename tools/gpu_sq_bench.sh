#!/bin/bash
# Developer tool (GPU box): SQ counter pass over the kernels of the bench step. Usage: tools/gpu_sq_bench.sh FILTER
set -e -o pipefail
FILT=${1:-lmd}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sqb_$FILT
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES \
  -d "$OUT/p1" -o sq --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-config5 --no-cpu-baseline > "$OUT/p1.log" 2> "$OUT/p1.err"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM \
  -d "$OUT/p2" -o sq --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-config5 --no-cpu-baseline > "$OUT/p2.log" 2> "$OUT/p2.err"
python3 - "$FILT" "$OUT/p1/sq_counter_collection.csv" "$OUT/p2/sq_counter_collection.csv" <<'PY'
import csv, sys, collections
filt = sys.argv[1]
for f in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{k:50s} {c:22s} {sum(v)/len(v):16.0f}")
PY
