#!/bin/bash
# round-4 visit F: chain walk of the clock re-run lanes (one round instead of two), on the round-3 framer kernels
mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4f_pytest.log 2>&1; tail -4 gpurun_out/r4f_pytest.log
run() { # tag, env (comma separated), flags
  echo "=== $1 | $2 | $3"
  ( env ${2//,/ } timeout 300 python bench.py --quick $3 ) > gpurun_out/r4f_$1.log 2>&1
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/r4f_$1.log | head -4 | tr '\n' ' '; echo
  python tools/stage_table.py gpurun_out/r4f_$1.log 2>/dev/null | sed -n '2p;$p'
  grep -i "error\|Traceback" gpurun_out/r4f_$1.log | head -3
}
F="--steps 20 --warmup 3"
WMBUS_DEBUG_ROUNDS=1 timeout 300 python bench.py --quick --steps 2 --warmup 1 2>&1 | grep "^rounds" | sort | uniq -c | sort -rn | head -12
for rep in 1 2; do
  run nochain_$rep  WMBUS_CLK_CHAINS=0 "$F"
  run chain_$rep    A=1 "$F"
  run chain1_$rep   WMBUS_FR_ROUNDS=1 "$F"
  run tnochain_$rep WMBUS_CLK_CHAINS=0 "$F --tolerance-mode"
  run tchain_$rep   A=1 "$F --tolerance-mode"
  run tchain1_$rep  WMBUS_FR_ROUNDS=1 "$F --tolerance-mode"
  run tc1k128_$rep  WMBUS_FR_ROUNDS=1,WMBUS_K3_BLOCKS=128 "$F --tolerance-mode"
  run tc1c10_$rep   WMBUS_FR_ROUNDS=1 "$F --tolerance-mode --contexts 10"
  run tc1c16_$rep   WMBUS_FR_ROUNDS=1 "$F --tolerance-mode --contexts 16"
done
