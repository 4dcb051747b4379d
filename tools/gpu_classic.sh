#!/bin/bash
# without SPLINES_VVISC / SPLINES_VDIFF on the GPU: kernels, 100-step runs, tiling
set -o pipefail
python -m pytest tests/test_gpu_classic_vertical.py -q -x -m gpu > gpurun_out/classic_tests.log 2>&1
echo "classic tests rc=$?" >> gpurun_out/classic_tests.log
tail -5 gpurun_out/classic_tests.log
python -m pytest tests/test_gpu_multitile.py -q -x -m gpu -k "classic" > gpurun_out/classic_mt.log 2>&1
echo "classic multitile rc=$?" >> gpurun_out/classic_mt.log
tail -3 gpurun_out/classic_mt.log
