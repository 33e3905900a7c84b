#!/bin/bash
export TMPDIR=/tmp
for c in "$@"; do
  echo "=== $c"
  ( timeout 600 python bench.py --no-cpu-baseline --no-check $c ) > gpurun_out/ctx.log 2>&1
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/ctx.log | tr '\n' ' '; grep -o 'host_decode_ms": [0-9.]*' gpurun_out/ctx.log | head -2 | tr '\n' ' '; echo; grep -i "error\|Traceback" gpurun_out/ctx.log | head -3
done
