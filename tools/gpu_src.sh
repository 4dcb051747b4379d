#!/bin/bash
# point sources on the GPU: kernels, whole runs, known answers, error rules, tiling
set -o pipefail
python -m pytest tests/test_gpu_sources.py tests/test_gpu_errors.py -q -x -m gpu > gpurun_out/src_tests.log 2>&1
echo "src tests rc=$?" >> gpurun_out/src_tests.log
tail -5 gpurun_out/src_tests.log
python -m pytest tests/test_gpu_multitile.py -q -x -m gpu -k "river" > gpurun_out/src_mt.log 2>&1
echo "src multitile rc=$?" >> gpurun_out/src_mt.log
tail -3 gpurun_out/src_mt.log
