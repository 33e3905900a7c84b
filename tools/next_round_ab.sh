#!/bin/bash
# First GPU visit of the next round, prepared at the end of round 1 (whose last changes could not be measured):
#   1. HERE (needs .git):   tools/next_round_ab.sh prepare      -> builds the "before" and candidate libraries
#   2. gpurun -- './tools/next_round_ab.sh run'                 -> parity suite, then in-box A/Bs
# Candidates:
#   r1gpu   main as of the last GPU-measured commit of round 1 (bad0c36: run-length framer v2)
#   (default) the working tree: + K1 job rotation, k3_bursts / k3_scan with loads in flight, burst_need fix
#   rlav3   branch next/rla-v3 (per-lane walk, k2_deglitch)
#   fb4     working tree with -DWM_FUSED_WAVES_PER_SIMD=4 (fused launch at 128 VGPRs)
#   lean    working tree with -DWM_FUSED_LEAN_CLOCK=1
# Runtime switches worth a row each: WMBUS_FUSE_FRAMERS=0, WMBUS_K3_BLOCKS=128/512.
set -e
cd "$(dirname "$0")/.."
L=$PWD/rtl-wmbus_amd
case "$1" in
prepare)
    make -s -C rtl-wmbus_amd
    ./tools/build_commit.sh r1gpu bad0c36
    ./tools/build_commit.sh rlav3 next/rla-v3
    ./tools/build_variant.sh fb4 -DWM_FUSED_WAVES_PER_SIMD=4
    ./tools/build_variant.sh lean -DWM_FUSED_LEAN_CLOCK=1
    ;;
run)
    export TMPDIR=/tmp
    mkdir -p gpurun_out/next
    ( timeout 420 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) | tee gpurun_out/next/suite.txt
    F="--steps 10 --warmup 2"
    ./tools/gpu_env.sh "WMBUS_HIP_LIB=$L/libwmbus_hip_r1gpu.so -- $F" "A=1 -- $F" "WMBUS_HIP_LIB=$L/libwmbus_hip_rlav3.so -- $F" \
        "WMBUS_HIP_LIB=$L/libwmbus_hip_fb4.so -- $F" "WMBUS_HIP_LIB=$L/libwmbus_hip_lean.so -- $F" "WMBUS_FUSE_FRAMERS=0 -- $F" \
        "WMBUS_HIP_LIB=$L/libwmbus_hip_rlav3.so,WMBUS_FUSE_FRAMERS=0 -- $F" "WMBUS_HIP_LIB=$L/libwmbus_hip_r1gpu.so -- $F" "A=1 -- $F" 2>&1 | tee gpurun_out/next/ab.txt
    # the parked run-length framer must pass the suite before it may replace the current one
    ( WMBUS_HIP_LIB=$L/libwmbus_hip_rlav3.so timeout 420 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) | tee gpurun_out/next/suite_rlav3.txt
    ;;
*) echo "usage: $0 prepare | run"; exit 2;;
esac
