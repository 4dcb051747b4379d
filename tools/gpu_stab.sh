#!/bin/bash
# TS_MIX_STABILITY on the GPU: the mixing kernels (plain and STAB instantiations), the golden fixture, 100-step runs
set -o pipefail
python -m pytest tests/test_gpu_biharmonic.py tests/test_golden.py -q -x -m gpu > gpurun_out/stab_tests.log 2>&1
echo "stab tests rc=$?" >> gpurun_out/stab_tests.log
tail -4 gpurun_out/stab_tests.log
