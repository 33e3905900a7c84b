#!/bin/bash
# round-4 visit E: does the packing of the (now small) framer blocks onto few CUs explain the longer stages?  LDS padding A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/rtl-wmbus_amd
run() { # tag, env (comma separated), flags
  echo "=== $1 | $2 | $3"
  ( env ${2//,/ } timeout 300 python bench.py --quick $3 ) > gpurun_out/r4e_$1.log 2>&1
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/r4e_$1.log | head -4 | tr '\n' ' '; echo
  python tools/stage_table.py gpurun_out/r4e_$1.log 2>/dev/null | sed -n '2p;$p'
  grep -i "error\|Traceback" gpurun_out/r4e_$1.log | head -3
}
F="--steps 20 --warmup 3"
for rep in 1 2; do
  run old_$rep    WMBUS_HIP_LIB=$L/libwmbus_hip_r4a.so,WMBUS_K1_STREAM=0 "$F"
  run new_$rep    A=1 "$F"
  run clk46_$rep  WMBUS_CLK_PAD=47000 "$F"
  run clk64_$rep  WMBUS_CLK_PAD=65000 "$F"
  run both_$rep   WMBUS_CLK_PAD=47000,WMBUS_RLA_PAD=14000 "$F"
  run rla14_$rep  WMBUS_RLA_PAD=14000 "$F"
  run rla22_$rep  WMBUS_CLK_PAD=47000,WMBUS_RLA_PAD=22000 "$F"
done
