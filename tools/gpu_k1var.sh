#!/bin/bash
# K1 duration of experiment builds (first launch only; downstream stages may fail on garbage)
mkdir -p gpurun_out/trace; export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
for l in "$@"; do
  WMBUS_HIP_LIB=$R/rtl-wmbus_amd/$l timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/trace -o v --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --contexts 1 --no-cpu-baseline --no-check > $R/gpurun_out/trace/logv.txt 2>&1
  python3 - "$l" <<'PY'
import csv,os,sys
R=os.environ['GRAFT_REPO_ROOT']
rows=[r for r in csv.DictReader(open(R+'/gpurun_out/trace/v_kernel_trace.csv')) if 'k1_demod' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows]
print(sys.argv[1], ['%.2f'%x for x in d])
PY
done
