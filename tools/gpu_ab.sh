#!/bin/bash
# in-box A/B of bench.py --quick over environment / flag sets:  tools/gpu_ab.sh TAG "ENV=V[,ENV=V] -- flags" ...   (every set twice, interleaved)
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; shift
for rep in 1 2; do
  i=0
  for c in "$@"; do
    i=$((i+1)); envs=${c%% -- *}; flags=${c#* -- }
    echo "=== $rep.$i | $envs | $flags"
    ( env ${envs//,/ } timeout 300 python bench.py --quick $flags ) > gpurun_out/${tag}_${i}_$rep.log 2>&1
    grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/${tag}_${i}_$rep.log | head -4 | tr '\n' ' '; echo
    python tools/stage_table.py gpurun_out/${tag}_${i}_$rep.log 2>/dev/null | sed -n '$p'
    grep -i "error\|Traceback" gpurun_out/${tag}_${i}_$rep.log | head -3
  done
done
