#!/bin/bash
# bench A/B over environment settings: each arg "ENV=VAL[,ENV=VAL] -- bench flags"
export TMPDIR=/tmp
for c in "$@"; do
  envs=${c%% -- *}; flags=${c#* -- }
  echo "=== $envs | $flags"
  ( env ${envs//,/ } timeout 600 python bench.py --no-cpu-baseline --no-check $flags ) > gpurun_out/env.log 2>&1
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|overlap: avg [0-9.]* ms' gpurun_out/env.log | tr '\n' ' '; echo; grep -i "error\|Traceback" gpurun_out/env.log | head -3
  python tools/stage_table.py gpurun_out/env.log 2>/dev/null | sed -n '2p;$p'
done
