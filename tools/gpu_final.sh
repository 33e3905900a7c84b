#!/bin/bash
# one-shot validation (used with 2 GPU-minutes left at the end of round 1): suite on the default build, A/B of
# the candidate builds, then the suite on the most ambitious candidate
export TMPDIR=/tmp
L=$PWD/rtl-wmbus_amd; F="--steps 10 --warmup 2"
mkdir -p gpurun_out/final
( timeout 45 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 ) | tee gpurun_out/final/suite_default.txt
./tools/gpu_env.sh "WMBUS_HIP_LIB=$L/libwmbus_hip_prev.so -- $F" "A=1 -- $F" "WMBUS_HIP_LIB=$L/libwmbus_hip_r2.so -- $F" "WMBUS_HIP_LIB=$L/libwmbus_hip_w5.so -- $F" 2>&1 | tee gpurun_out/final/ab.txt
( WMBUS_HIP_LIB=$L/libwmbus_hip_r2.so timeout 40 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 ) | tee gpurun_out/final/suite_r2.txt
