#!/bin/bash
# Developer tool (GPU box): the cost of the RCCL exchanges in loopback -- bench line with and without, kernel statistics
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-BENCHMARK1}
OUT=$R/gpurun_out/loopback_$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$R
cd "$R"
python3 bench.py --config $CFG --steps 40 --warmup 5 --no-config5 --no-cpu-baseline > "$OUT/plain.json" 2> "$OUT/plain.err"
python3 bench.py --config $CFG --steps 40 --warmup 5 --no-config5 --no-cpu-baseline --loopback > "$OUT/loop.json" 2> "$OUT/loop.err"
python3 - "$OUT" <<'PY'
import json, sys
o = sys.argv[1]
a = json.loads(open(o + "/plain.json").read().strip().splitlines()[-1]); b = json.loads(open(o + "/loop.json").read().strip().splitlines()[-1])
print("plain ms/step", a["ms_per_step"], "loopback ms/step", b["ms_per_step"], "diff", b["ms_per_step"] - a["ms_per_step"])
print("step2d_loop", a["kernel_ms"]["step2d_loop"], b["kernel_ms"]["step2d_loop"])
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o lb --output-format csv -- python3 "$R/bench.py" --config $CFG --steps 10 --warmup 2 --no-config5 --no-cpu-baseline --loopback > "$OUT/prof.json" 2> "$OUT/prof.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/**/lb_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if float(r["Percentage"]) > 1.0:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Percentage"]}%')
PY
