#!/bin/bash
# bench A/B over flag sets, inside one box visit: each arg "ENV=VAL[,ENV=VAL] -- bench flags" (A=1 for no env); prints value per arg, two rounds
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2; do
  for c in "$@"; do
    envs=${c%% -- *}; flags=${c#* -- }
    v=$( ( env ${envs//,/ } timeout 300 python bench.py --no-cpu-baseline --no-check --steps 10 --warmup 2 $flags ) 2>&1 | tail -1 | grep -o '"value": [0-9.]*' | head -1 )
    echo "rep$rep | $envs | $flags | $v"
  done
done
