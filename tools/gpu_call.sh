#!/bin/bash
# one GPU-box visit: parity tests, bench line (args: extra bench flags)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log
( time timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" ) > gpurun_out/bench.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"roofline": {[^}]*}\|"parity_check": "[^"]*"\|"stage_ms_last_step": \[{[^}]*}' gpurun_out/bench.log
tail -4 gpurun_out/bench.log | cut -c1-300
