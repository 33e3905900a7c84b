#!/bin/bash
# profiles for profiles/: (1) kernel trace + stats of the default bench command, (2) FETCH_SIZE and
# WRITE_SIZE PMC passes (separate runs, single context so that one k1 launch covers all captures)
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 1 > $R/gpurun_out/prof/bench_under_rocprof.json 2> $R/gpurun_out/prof/trace.log
tail -1 $R/gpurun_out/prof/bench_under_rocprof.json | cut -c1-400
# (1b) the same kernels un-overlapped: one context of 1024 captures, nothing runs beside anything
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace1 -o single --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-check > $R/gpurun_out/prof/single_context_under_rocprof.json 2> $R/gpurun_out/prof/trace1.log
tail -1 $R/gpurun_out/prof/single_context_under_rocprof.json | cut -c1-300
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --stats -d $R/gpurun_out/prof/pmc_$ctr -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --contexts 1 --no-cpu-baseline --no-check > $R/gpurun_out/prof/pmc_$ctr.log 2>&1
  echo "$ctr rc=$?"
done
python3 - <<'PY'
import csv,os,collections,json
R=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof'
out={}
for ctr in ('FETCH_SIZE','WRITE_SIZE'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f'{R}/pmc_{ctr}/pmc_counter_collection.csv')):
        if r['Counter_Name']==ctr: agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    with open(f'{R}/pmc_{ctr.lower()}.csv','w') as f:
        f.write('kernel,counter,launches,sum_KB,avg_KB\n')
        for k,v in sorted(agg.items()): f.write(f'"{k}",{ctr},{len(v)},{sum(v):.1f},{sum(v)/len(v):.1f}\n')
    out[ctr]={k:(len(v),sum(v)) for k,v in agg.items()}
json.dump(out,open(f'{R}/pmc_summary.json','w'),indent=1)
for ctr,d in out.items():
    for k,(n,sv) in d.items():
        if any(x in k for x in ('k1_','k2_','k3_')): print(ctr,k,n,round(sv/1e6,3),'GB')
PY
