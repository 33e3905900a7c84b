#!/bin/bash
# bench.py --quick over sets of extra flags, two interleaved rounds, with the per-context stage means:  tools/tune_sweep.sh "" "--tune k1_tiles_per_block=4" ...
cfgs=("$@")
for rep in 1 2; do
for cfg in "${cfgs[@]}"; do
  python bench.py --quick --steps 10 --warmup 2 $cfg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
dd=json.load(open(d['details'])); import statistics as st
rows=dd['stage_ms_mid_step']; m=lambda k: round(st.mean(r[k] for r in rows),2)
print('rep $rep [$cfg]:', d['value'], d['ms_per_step'], 'K1', m('demod_ms'), 'alone', d['roofline']['alone']['avg_launch_ms'], 'clock', m('clock_ms'), 'rla', m('rla_ms'), 'gather', m('gather_ms'), 'turn', m('turn_wait_ms'), 'chain', m('gpu_total_ms'))"
done; done
