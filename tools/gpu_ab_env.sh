#!/bin/bash
# in-box A/B: each argument "ENV=VAL[,ENV=VAL]" (use A=1 for the default); every configuration twice, interleaved
cd "$(dirname "$0")/.."
F="${WMBUS_AB_FLAGS:---steps 10 --warmup 2}"
args=()
for rep in 1 2; do for e in "$@"; do args+=("$e -- $F"); done; done
./tools/gpu_env.sh "${args[@]}" 2>&1 | grep -v "turn_wait  "
