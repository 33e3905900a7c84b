#!/bin/bash
# tolerance-mode sweep: clock segment length x contexts, two interleaved rounds
for rep in 1 2; do
cfgs=("0 0" "16384 0" "16384 12" "32768 12" "16384 16" "65536 0")
for cfg in "${cfgs[@]}"; do
  set -- $cfg
  v=$(python bench.py --quick --tolerance-mode --steps 10 --warmup 2 --seg-len $1 --contexts $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "rep $rep seg $1 contexts $2: $v"
done; done
