#!/bin/bash
# Developer tool (GPU box): MPDATA parity tests, then rocprofv3 kernel statistics of the tracer step of configuration 5.
# Usage: tools/gpu_mp.sh TAG [notest]
set -o pipefail
TAG=${1:-mp}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/mp_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$R
cd "$R"
if [ "$2" != notest ]; then
  python3 -m pytest tests/test_gpu_mpdata.py tests/test_basin.py tests/test_masking.py -m gpu -x -q -k "mpdata or MPDATA" > "$OUT/tests.log" 2>&1
  tail -4 "$OUT/tests.log"
fi
python3 tools/bench_mpdata.py BENCHMARK3 5 > "$OUT/bm.log" 2>&1; tail -2 "$OUT/bm.log"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o mp --output-format csv -- python3 "$R/tools/bench_mpdata.py" BENCHMARK3 3 > "$OUT/prof.log" 2> "$OUT/prof.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/**/mp_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if float(r["Percentage"]) > 0.5:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
PY
