#!/bin/bash
# the tracer step: its tests, then a short bench (driver line of step3d_t)
set -o pipefail
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_main3d.py tests/test_masking.py tests/test_basin.py tests/test_golden.py tests/test_gpu_sources.py -q -x -m gpu -k "step3d_t or steps or main3d or levels or tracer or golden or 100" > gpurun_out/t_tests.log 2>&1
echo "t tests rc=$?" >> gpurun_out/t_tests.log
tail -3 gpurun_out/t_tests.log
for r in 1 2; do
python bench.py --steps 30 --warmup 5 --no-config5 --no-cpu-baseline > gpurun_out/bench_t.json 2> gpurun_out/bench_t.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_t.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "step3d_t", d["kernel_ms"]["step3d_t"], "frac", d["roofline"]["frac"], "avg_ms", d["roofline"]["avg_ms"])
PY
done
