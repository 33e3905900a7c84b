#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for m in od full; do
  [ $m = full ] && export WMBUS_RSSI_FULL=1 || unset WMBUS_RSSI_FULL
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/odp_$m -o s --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --contexts 1 --quick > /dev/null 2>&1
  echo "== $m single context"; f=$(find $R/gpurun_out/odp_$m -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | head -9
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/odq_$m -o s --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --quick > /dev/null 2>&1
  echo "== $m 8 contexts"; f=$(find $R/gpurun_out/odq_$m -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | head -11
done
