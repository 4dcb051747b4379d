#!/bin/bash
# MY25_MIXING (and the GLS closure beside it) on the GPU: kernels, reference vectors, 100-step runs, tiling
set -o pipefail
python -m pytest tests/test_gpu_gls.py tests/test_golden.py -q -x -m gpu -k "gls or my25" > gpurun_out/my25_tests.log 2>&1
echo "my25 tests rc=$?" >> gpurun_out/my25_tests.log
tail -5 gpurun_out/my25_tests.log
python -m pytest tests/test_gpu_multitile.py -q -x -m gpu -k "my25 or gls" > gpurun_out/my25_mt.log 2>&1
echo "my25 multitile rc=$?" >> gpurun_out/my25_mt.log
tail -3 gpurun_out/my25_mt.log
