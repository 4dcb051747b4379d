#!/bin/bash
# the whole -m gpu suite, progress to a file (gpurun takes silence for a hang)
set -o pipefail
python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1
echo "gpu suite rc=$?" >> gpurun_out/gpu_suite.log
tail -15 gpurun_out/gpu_suite.log
