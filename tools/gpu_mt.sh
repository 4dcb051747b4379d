#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_multitile.py tests/test_gpu_rccl.py tests/test_gpu_fortran_host.py -q -x -m gpu > gpurun_out/mt_tests.log 2>&1
echo "mt tests rc=$?" >> gpurun_out/mt_tests.log
tail -4 gpurun_out/mt_tests.log
bash tools/gpu_loopback.sh BENCHMARK1 2>&1 | head -4
