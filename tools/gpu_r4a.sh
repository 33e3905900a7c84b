#!/bin/bash
# round-4 visit A: parity tests (default + side stream), A/B of the K1 stream / side stream / SR window / EMA warm-up, the new bench legs
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/rtl-wmbus_amd
( time timeout 600 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4a_pytest.log 2>&1; tail -4 gpurun_out/r4a_pytest.log
( time WMBUS_RLA_SIDE=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q ) > gpurun_out/r4a_pytest_side.log 2>&1; tail -3 gpurun_out/r4a_pytest_side.log
run() { # tag, env (comma separated), flags
  echo "=== $1 | $2 | $3"
  ( env ${2//,/ } timeout 300 python bench.py --quick $3 ) > gpurun_out/r4a_$1.log 2>&1
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/r4a_$1.log | head -4 | tr '\n' ' '; echo
  python tools/stage_table.py gpurun_out/r4a_$1.log 2>/dev/null | sed -n '2p;$p'
  grep -i "error\|Traceback" gpurun_out/r4a_$1.log | head -3
}
F="--steps 20 --warmup 3"
for rep in 1 2; do
  run k1ev_$rep   WMBUS_K1_STREAM=0 "$F"
  run k1st_$rep   A=1 "$F"
  run side_$rep   WMBUS_RLA_SIDE=1 "$F"
  run sr0_$rep    WMBUS_HIP_LIB=$L/libwmbus_hip_sr0.so "$F"
  run ema24_$rep  WMBUS_HIP_LIB=$L/libwmbus_hip_ema24.so "$F"
  run tol12_$rep  A=1 "$F --tolerance-mode"
  run tol12ns_$rep WMBUS_RLA_SIDE=0 "$F --tolerance-mode"
  run tol8_$rep   A=1 "$F --tolerance-mode --contexts 8"
  run tol10_$rep  GPU_MAX_HW_QUEUES=24 "$F --tolerance-mode --contexts 10"
  run tol12q_$rep GPU_MAX_HW_QUEUES=24 "$F --tolerance-mode"
done
# the full bench line with every leg (what the driver runs)
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4a_bench_full.json 2> gpurun_out/r4a_bench_full.err
tail -c 6000 gpurun_out/r4a_bench_full.json; tail -5 gpurun_out/r4a_bench_full.err
