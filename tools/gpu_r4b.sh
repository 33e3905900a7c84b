#!/bin/bash
# round-4 visit B: the wave-transposed soft symbols / slicer words and the register-resident clock kernel against the r4a build
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/rtl-wmbus_amd
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4b_pytest.log 2>&1; tail -4 gpurun_out/r4b_pytest.log
run() { # tag, env (comma separated), flags
  echo "=== $1 | $2 | $3"
  ( env ${2//,/ } timeout 300 python bench.py --quick $3 ) > gpurun_out/r4b_$1.log 2>&1
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' gpurun_out/r4b_$1.log | head -4 | tr '\n' ' '; echo
  python tools/stage_table.py gpurun_out/r4b_$1.log 2>/dev/null | sed -n '2p;$p'
  grep -i "error\|Traceback" gpurun_out/r4b_$1.log | head -3
}
F="--steps 20 --warmup 3"
for rep in 1 2; do
  run old_$rep    WMBUS_HIP_LIB=$L/libwmbus_hip_r4a.so,WMBUS_K1_STREAM=0 "$F"
  run new_$rep    A=1 "$F"
  run wpb2_$rep   WMBUS_HIP_LIB=$L/libwmbus_hip_wpb2.so "$F"
  run wpb8_$rep   WMBUS_HIP_LIB=$L/libwmbus_hip_wpb8.so "$F"
  run c10_$rep    A=1 "$F --contexts 10"
  run c12_$rep    A=1 "$F --contexts 12"
  run seg64_$rep  A=1 "$F --seg-len 65536"
  run s1span_$rep WMBUS_S1_SPAN=2 "$F"
  run tolold_$rep WMBUS_HIP_LIB=$L/libwmbus_hip_r4a.so,WMBUS_K1_STREAM=0,WMBUS_RLA_SIDE=0 "$F --tolerance-mode"
  run tol12_$rep  A=1 "$F --tolerance-mode"
  run tol16_$rep  A=1 "$F --tolerance-mode --contexts 16"
done
run single1 A=1 "--steps 3 --warmup 1 --contexts 1"
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4b_bench_full.json 2> gpurun_out/r4b_bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b_bench_full.json').read().strip().splitlines()[-1])
print('FULL', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['in_timed_region']['avg_launch_ms'], d['parity']['ok'])
print('tol', d['tolerance_mode_leg'].get('value'), d['tolerance_mode_leg'].get('differing_lines'), d['tolerance_mode_leg'].get('last_pass'))
for k in ('c2_single_stream','c3_single_stream','c3_batch','cli'):
    print(k, {kk:vv for kk,vv in d.get(k,{}).items() if kk not in ('workload','runs','command')})
PY
tail -3 gpurun_out/r4b_bench_full.err
