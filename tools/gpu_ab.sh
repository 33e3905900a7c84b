#!/bin/bash
# A/B: single-context bench runs (kernel times un-overlapped) for env-selected variants
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "$@"; do
  echo "=== $v"
  ( env $v timeout 600 python bench.py --steps 2 --warmup 1 --contexts 1 --streams ${STREAMS:-256} --no-cpu-baseline --no-check ) > gpurun_out/ab.log 2>&1
  grep -o '"stage_ms_last_step": \[{[^}]*}' gpurun_out/ab.log || tail -5 gpurun_out/ab.log
done
