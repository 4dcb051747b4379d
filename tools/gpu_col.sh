#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_main3d.py tests/test_masking.py tests/test_basin.py -q -x -m gpu -k "pre_step3d or step3d_uv or rhs3d or rhs_pieces or steps or main3d or levels or uv3dmix" > gpurun_out/col_tests.log 2>&1
echo "col tests rc=$?" >> gpurun_out/col_tests.log
tail -3 gpurun_out/col_tests.log
python bench.py --steps 20 --warmup 5 --no-config5 --no-cpu-baseline > gpurun_out/bench_col.json 2> gpurun_out/bench_col.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_col.json').read().strip().splitlines()[-1])
k=d["kernel_ms"]
print("ms_per_step", d["ms_per_step"], "pre_step3d", k["pre_step3d"], "step3d_uv", k["step3d_uv"], "step2d_loop", k["step2d_loop"], "lmd", k["lmd_vmix"], "step3d_t", k["step3d_t"])
PY
