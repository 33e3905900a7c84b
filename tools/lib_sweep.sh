#!/bin/bash
# bench.py --quick for several builds of the library (tools/build_variant.sh), two interleaved rounds:  tools/lib_sweep.sh - sleep5 sleep10 [-- bench flags]
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in 1 2; do
for l in "${libs[@]}"; do
  if [ "$l" == "-" ]; then unset WMBUS_HIP_LIB; else export WMBUS_HIP_LIB=$PWD/rtl-wmbus_amd/libwmbus_hip_$l.so; fi
  python bench.py --quick --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
dd=json.load(open(d['details'])); import statistics as st
rows=dd['stage_ms_mid_step']; m=lambda k: round(st.mean(r[k] for r in rows),2)
print('rep $rep lib $l:', d['value'], d['ms_per_step'], 'K1', m('demod_ms'), 'clock', m('clock_ms'), 'rla', m('rla_ms'), 'gather', m('gather_ms'), 'turn', m('turn_wait_ms'), 'chain', m('gpu_total_ms'))"
done; done
