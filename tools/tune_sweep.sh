#!/bin/bash
# bench.py --quick over sets of extra flags, two interleaved rounds:  tools/tune_sweep.sh "" "--tune k1_tiles_per_block=4" ...
cfgs=("$@")
for rep in 1 2; do
for cfg in "${cfgs[@]}"; do
  v=$(python bench.py --quick --steps 10 --warmup 2 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "rep $rep [$cfg]: $v"
done; done
