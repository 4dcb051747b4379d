#!/bin/bash
# WET_DRY on the GPU: the new tests, then the tests of the files the change touched
set -o pipefail
python -m pytest tests/test_gpu_wetdry.py tests/test_wetdry.py tests/test_golden.py -q -x -m gpu -k "wet" > gpurun_out/wet_tests.log 2>&1
echo "wet tests rc=$?" >> gpurun_out/wet_tests.log
tail -30 gpurun_out/wet_tests.log
