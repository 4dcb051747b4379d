#!/bin/bash
set -o pipefail
python -m pytest tests/ -q -x -m gpu -k "lmd or kpp or physics or smoke" > gpurun_out/lmd_tests.log 2>&1
echo "lmd tests rc=$?" >> gpurun_out/lmd_tests.log
tail -5 gpurun_out/lmd_tests.log
python bench.py --steps 20 --warmup 5 --no-config5 > gpurun_out/bench_lmd.json 2> gpurun_out/bench_lmd.err
tail -c 3000 gpurun_out/bench_lmd.json
