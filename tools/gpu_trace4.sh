#!/bin/bash
# kernel trace of the default 4-context bench: union busy time, overlap, per-kernel totals for the last step
mkdir -p gpurun_out/trace4; export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace4 -o t --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check > $R/gpurun_out/trace4/log.txt 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $R/gpurun_out/trace4/log.txt | tr '\n' ' '; echo
python3 - <<'PY'
import csv,os,collections
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/trace4/t_kernel_trace.csv')))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-28:],int(r['Grid_Size_X'])) for r in rows]
ev.sort()
# last step = after the 3rd-from-last batch of k1_demod launches: take window covering the last 4 k1 launches to the end
k1=[e for e in ev if 'k1_demod' in e[2]]
t0=k1[-4][0]; t1=max(e[1] for e in ev)
win=[e for e in ev if e[0]>=t0]
span=(t1-t0)/1e6
# union busy
cur=None; busy=0
for s,e,_,_ in sorted(win):
    if cur is None: cur=[s,e]
    elif s<=cur[1]: cur[1]=max(cur[1],e)
    else: busy+=cur[1]-cur[0]; cur=[s,e]
busy+=cur[1]-cur[0]
print(f"last step window {span:.2f} ms, union busy {busy/1e6:.2f} ms")
tot=collections.Counter(); cnt=collections.Counter()
for s,e,n,g in win: tot[n]+=e-s; cnt[n]+=1
for n,v in tot.most_common(8): print(f"  {n:30s} sum {v/1e6:8.2f} ms  n {cnt[n]}")
# timeline of big kernels
for s,e,n,g in win:
    if e-s>300000: print(f"   {(s-t0)/1e6:8.2f} -> {(e-t0)/1e6:8.2f}  {n[-22:]:22s} grid {g}")
PY
