#!/bin/bash
# profiles for profiles/: (1) kernel trace + stats of the default bench command, (2) FETCH_SIZE and
# WRITE_SIZE PMC passes (separate runs, single context so that one k1 launch covers all captures)
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
[ -n "$PMC_ONLY" ] || timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-tolerance-leg --no-legs > $R/gpurun_out/prof/bench_under_rocprof.json 2> $R/gpurun_out/prof/trace.log
tail -1 $R/gpurun_out/prof/bench_under_rocprof.json | cut -c1-400
# (1b) the same kernels un-overlapped: one context of 1024 captures, nothing runs beside anything
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace1 -o single --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --contexts 1 --quick > $R/gpurun_out/prof/single_context_under_rocprof.json 2> $R/gpurun_out/prof/trace1.log
tail -1 $R/gpurun_out/prof/single_context_under_rocprof.json | cut -c1-300
# (1c) configs[2] at batch size: the -d 5 -s instantiation of the demodulation kernel and the framers behind it
[ -n "$PMC_ONLY" ] || timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace_c3 -o c3 --output-format csv -- python $R/tools/gpu_c3.py > $R/gpurun_out/prof/c3_under_rocprof.json 2> $R/gpurun_out/prof/trace_c3.log
tail -1 $R/gpurun_out/prof/c3_under_rocprof.json | cut -c1-300
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --stats -d $R/gpurun_out/prof/pmc_$ctr -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --contexts 1 --quick > $R/gpurun_out/prof/pmc_$ctr.log 2>&1
  echo "$ctr rc=$?"
done
# (3) SQ counters of the same single-context run (1024 captures per launch), each group in its own pass: what the VALU
# roofline of the bench line is computed from (instruction counts) and what the waves spend their time on
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --stats -d $R/gpurun_out/prof/pmc_sq$i -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --contexts 1 --quick > $R/gpurun_out/prof/pmc_sq$i.log 2>&1
  echo "SQ group $i rc=$?"
done
python3 - <<'PY'
import csv,os,collections,json
R=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof'
out={}
sq=collections.defaultdict(lambda: collections.defaultdict(list)); durs=collections.defaultdict(list)
for i in (1,2,3):
    p=f'{R}/pmc_sq{i}/pmc_counter_collection.csv'
    if not os.path.exists(p): continue
    seen=set()
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        sq[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if i==1 and r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); durs[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
with open(f'{R}/pmc_sq.csv','w') as f:
    f.write('kernel,launches,avg_ms_under_pmc,counter,sum,avg_per_launch\n')
    for k,d in sorted(sq.items()):
        for c,v in sorted(d.items()):
            f.write(f'"{k}",{len(v)},{(sum(durs[k])/max(1,len(durs[k]))):.4f},{c},{sum(v):.0f},{sum(v)/len(v):.0f}\n')
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
