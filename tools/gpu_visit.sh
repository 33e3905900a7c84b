#!/bin/bash
# One GPU-box visit of round 5: GPU tests, the K1 stage stamps, A/B of library builds / tuning fields, one full bench run.
#   tools/gpu_visit.sh TAG [tests] [stamps] [ab "LIB|flags" ...] [full]
# A/B sets: "LIBFILE|bench flags" (LIBFILE = - for the working tree's libwmbus_hip.so), every set twice, interleaved.
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
R=$PWD
abs=()
for a in "$@"; do
  case "$a" in
    tests) ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/pytest_gpu.log 2>&1; tail -6 $out/pytest_gpu.log ;;
    stamps) WMBUS_HIP_LIB=$R/rtl-wmbus_amd/libwmbus_hip_stamps.so timeout 600 python tools/gpu_k1_stamps.py --ring > $out/k1_stage_cycles.txt 2> $out/k1_stage_cycles.err; cat $out/k1_stage_cycles.txt; tail -3 $out/k1_stage_cycles.err ;;
    full) ( time timeout 900 python bench.py ) > $out/bench_full.log 2>&1; tail -c 2500 $out/bench_full.log; cp gpurun_out/bench_details.json $out/bench_details.json 2>/dev/null ;;
    rehearse8)   # the eight-rank host side on ONE GPU (eight ranks x 128 captures over gloo on device 0) against one rank alone
      ( time timeout 600 python bench.py --gpus 1 --streams 128 --steps 10 --warmup 2 --no-cpu-baseline --no-legs --no-tolerance-leg --details $out/rehearse_1rank.json ) > $out/rehearse_1rank.log 2>&1
      ( time WMBUS_BENCH_DEVICE=0 WMBUS_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --streams 128 --steps 10 --warmup 2 --no-cpu-baseline --no-legs --no-tolerance-leg --details $out/rehearse_8ranks.json ) > $out/rehearse_8ranks.log 2>&1
      python - $out <<'PY'
import json, sys
o = sys.argv[1]
for name in ("rehearse_1rank", "rehearse_8ranks"):
    try:
        d = json.load(open(f"{o}/{name}.json"))
        print(name, "value", d["value"], "ms_per_step", d["ms_per_step"], "setup", d["setup_s"], "host", d["host"], "parity", d.get("parity_check"),
              "oracle_s", [d["parity"][k]["oracle_s"] for k in ("first_pass", "last_pass")] if "parity" in d else None)
    except Exception as e:
        print(name, "no record:", e)
PY
      tail -3 $out/rehearse_8ranks.log | cut -c1-600 ;;
    *) abs+=("$a") ;;
  esac
done
if [ ${#abs[@]} -gt 0 ]; then
  for rep in 1 2; do
    i=0
    for c in "${abs[@]}"; do
      i=$((i+1)); lib=${c%%|*}; flags=${c#*|}
      [ "$lib" = "-" ] && libenv="" || libenv="WMBUS_HIP_LIB=$R/rtl-wmbus_amd/$lib"
      ( env $libenv timeout 300 python bench.py --quick --steps 20 --warmup 3 --details $out/ab_${i}_$rep.json $flags ) > $out/ab_${i}_$rep.log 2>&1
      echo "=== $rep.$i | $lib | $flags | $(grep -o '"value":[0-9.]*\|"ms_per_step":[0-9.]*\|"avg_launch_ms":[0-9.]*\|"in_region_ms":[0-9.]*' $out/ab_${i}_$rep.log | head -4 | tr '\n' ' ')"
      grep -i "error\|Traceback" $out/ab_${i}_$rep.log | head -3
    done
  done
fi
