#!/bin/bash
# in-box A/B of library builds: tools/gpu_bisect.sh tag1 tag2 ... ("head" = the default build); every tag twice, interleaved
cd "$(dirname "$0")/.."
L=$PWD/rtl-wmbus_amd
F="--steps 10 --warmup 2"
args=()
for rep in 1 2; do
  for t in "$@"; do
    if [ "$t" = head ]; then args+=("A=1 -- $F"); else args+=("WMBUS_HIP_LIB=$L/libwmbus_hip_$t.so -- $F"); fi
  done
done
./tools/gpu_env.sh "${args[@]}"
