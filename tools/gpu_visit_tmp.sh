#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/v_tests.log 2>&1
cat gpurun_out/v_tests.log
B=$GRAFT_REPO_ROOT/rtl-wmbus_amd/libwmbus_hip_before.so
./tools/gpu_ab.sh v "WMBUS_HIP_LIB=$B -- " "X=1 -- "
# instruction counters of both builds, single context
cd /tmp
for t in before after; do
  [ $t = before ] && export WMBUS_HIP_LIB=$B || unset WMBUS_HIP_LIB
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/v_sq_$t -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --contexts 1 --quick > $GRAFT_REPO_ROOT/gpurun_out/v_sq_$t.log 2>&1
  echo "sq $t rc=$?"
done
unset WMBUS_HIP_LIB
python3 - <<'PY'
import csv, os, collections
R = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out'
for t in ('before', 'after'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    p = None
    for root, _, files in os.walk(f'{R}/v_sq_{t}'):
        for f in files:
            if f.endswith('counter_collection.csv'): p = os.path.join(root, f)
    if not p: print(t, 'no counters'); continue
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
    for k, d in sorted(agg.items()):
        if 'k2_rla' in k or 'k1_' in k or 'k2_clock' in k:
            print(t, k, len(n[k]), ' '.join('%s=%.3fG' % (c.replace('SQ_', ''), v / 1e9) for c, v in sorted(d.items())))
PY
