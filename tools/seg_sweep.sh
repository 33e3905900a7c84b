#!/bin/bash
# exact-mode bench.py --quick over clock segment lengths x contexts, two interleaved rounds:  tools/seg_sweep.sh "SEG CTX" ...
[ $# -eq 0 ] && set -- "0 0" "131072 0" "131072 12" "65536 12"
cfgs=("$@")
for rep in 1 2; do
for cfg in "${cfgs[@]}"; do
  a=($cfg)
  v=$(python bench.py --quick --steps 10 --warmup 2 --seg-len ${a[0]} --contexts ${a[1]} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "rep $rep seg ${a[0]} contexts ${a[1]}: $v"
done; done
