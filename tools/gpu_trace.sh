#!/bin/bash
# kernel trace of one single-context bench run: per-dispatch durations; args = env assignments
mkdir -p gpurun_out/trace; export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace -o t --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --contexts 1 --streams ${STREAMS:-1024} --no-cpu-baseline --no-check > $R/gpurun_out/trace/log.txt 2>&1
python3 - <<'PY'
import csv,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/trace/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-34:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    if d>0.3: print(f"{(int(r['Start_Timestamp'])-t0)/1e6:10.3f} ms  dur {d:8.3f} ms  grid {r['Grid_Size_X']:>10s}  {r['Kernel_Name'][:40]}")
PY
