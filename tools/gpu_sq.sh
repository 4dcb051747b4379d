#!/bin/bash
# Developer tool (GPU box): SQ counter pass (where do the wave cycles go) for one entry.
# Usage: tools/gpu_sq.sh ENTRY [CONFIG] ["param=value ..."]
set -e -o pipefail
ENTRY=$1; CONFIG=${2:-BENCHMARK3}; SETS=${3:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq_$ENTRY$(echo "$SETS" | tr -c "a-zA-Z0-9\n" "_")
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES \
  -d "$OUT/p1" -o sq --output-format csv -- python3 "$R/tools/bench_kernel.py" $CONFIG $ENTRY 2 $SETS > "$OUT/p1.log" 2> "$OUT/p1.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  -d "$OUT/p2" -o sq --output-format csv -- python3 "$R/tools/bench_kernel.py" $CONFIG $ENTRY 2 $SETS > "$OUT/p2.log" 2> "$OUT/p2.err"
python3 - "$OUT/p1/sq_counter_collection.csv" "$OUT/p2/sq_counter_collection.csv" <<'PY'
import csv, sys, collections
for f in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if "rocclr" in k or "periodic" in k or "wall_bc" in k: continue
        print(f"{k:50s} {c:22s} {sum(v)/len(v):16.0f}")
PY
