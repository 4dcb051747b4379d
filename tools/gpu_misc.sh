#!/bin/bash
set -o pipefail
python -m pytest tests/test_golden.py tests/test_gpu_mpdata.py tests/test_gpu_errors.py tests/test_gpu_kernels.py -q -x -m gpu -k "atm or press_compensate or point_sources or flather" > gpurun_out/misc_tests.log 2>&1
echo "misc tests rc=$?" >> gpurun_out/misc_tests.log
tail -4 gpurun_out/misc_tests.log
