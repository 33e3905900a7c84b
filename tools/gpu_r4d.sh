#!/bin/bash
# round-4 visit D: register-resident clock kernel with a full block of lead, uniform run-length kernel (62 VGPRs), issue priority for the framers
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/rtl-wmbus_amd
./tools/clkbench | grep "blocks  256\|blocks    1 active 64"
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4d_pytest.log 2>&1; tail -4 gpurun_out/r4d_pytest.log
run() { # tag, env (comma separated), flags
  echo "=== $1 | $2 | $3"
  ( env ${2//,/ } timeout 300 python bench.py --quick $3 ) > gpurun_out/r4d_$1.log 2>&1
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/r4d_$1.log | head -4 | tr '\n' ' '; echo
  python tools/stage_table.py gpurun_out/r4d_$1.log 2>/dev/null | sed -n '2p;$p'
  grep -i "error\|Traceback" gpurun_out/r4d_$1.log | head -3
}
F="--steps 20 --warmup 3"
for rep in 1 2; do
  run old_$rep    WMBUS_HIP_LIB=$L/libwmbus_hip_r4a.so,WMBUS_K1_STREAM=0 "$F"
  run new_$rep    A=1 "$F"
  run prio1_$rep  WMBUS_HIP_LIB=$L/libwmbus_hip_prio1.so "$F"
  run prio3_$rep  WMBUS_HIP_LIB=$L/libwmbus_hip_prio3.so "$F"
  run p3c10_$rep  WMBUS_HIP_LIB=$L/libwmbus_hip_prio3.so "$F --contexts 10"
  run p3c12_$rep  WMBUS_HIP_LIB=$L/libwmbus_hip_prio3.so "$F --contexts 12"
  run c12_$rep    A=1 "$F --contexts 12"
  run tolold_$rep WMBUS_HIP_LIB=$L/libwmbus_hip_r4a.so,WMBUS_K1_STREAM=0,WMBUS_RLA_SIDE=0 "$F --tolerance-mode"
  run tol12_$rep  A=1 "$F --tolerance-mode"
  run tolp3_$rep  WMBUS_HIP_LIB=$L/libwmbus_hip_prio3.so "$F --tolerance-mode"
done
run single1 A=1 "--steps 3 --warmup 1 --contexts 1"
python tools/gpu_single.py 2>&1 | tail -8
