#!/bin/bash
export TMPDIR=/tmp
for c in "$@"; do
  echo "=== contexts $c"
  ( timeout 600 python bench.py --steps 4 --warmup 1 --contexts $c --no-cpu-baseline --no-check ) > gpurun_out/ctx.log 2>&1
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/ctx.log | tr '\n' ' '; echo
done
