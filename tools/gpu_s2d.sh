#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_main3d.py tests/test_masking.py -q -x -m gpu -k "step2d or steps or main3d" > gpurun_out/s2d_tests.log 2>&1
echo "s2d tests rc=$?" >> gpurun_out/s2d_tests.log
tail -3 gpurun_out/s2d_tests.log
python bench.py --steps 20 --warmup 5 --no-config5 --no-cpu-baseline > gpurun_out/bench_s2d.json 2> gpurun_out/bench_s2d.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_s2d.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "step2d_loop", d["kernel_ms"]["step2d_loop"], "lmd_vmix", d["kernel_ms"]["lmd_vmix"])
PY
