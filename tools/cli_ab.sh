#!/bin/bash
# Interleaved A/B of two library builds under the batch CLI: the tree's libwmbus_hip.so against a variant LD_PRELOADed over it
#   KEEP_FILES=1 CLI_SKIP_FROM_HOST=1 tools/bench_cli.sh > /dev/null; tools/cli_ab.sh [variant.so]; rm -rf /dev/shm/wmbus_cli_bench
D=/dev/shm/wmbus_cli_bench; FILES=$(ls $D/f*.cu8)
for i in 1 2 3; do
  V=${1:-$PWD/rtl-wmbus_amd/libwmbus_hip_perstream.so}
  for lib in "" "$V"; do
    LD_PRELOAD=$lib ./rtl-wmbus_amd/rtl_wmbus_hip -S -v $FILES > /dev/null 2> $D/err_ab.txt
    echo "$( [ -z "$lib" ] && echo tree || basename $lib ) $(grep total: $D/err_ab.txt | sed 's/.*decode/decode/')"
  done
done
