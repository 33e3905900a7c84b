#!/bin/bash
# kernel trace of a quick 8-context bench run: per-kernel stats + one context's timeline.  tools/gpu_trace_quick.sh TAG [bench flags]
tag=$1; shift
mkdir -p gpurun_out/$tag; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o t --output-format csv -- python $R/bench.py --quick --steps 8 --warmup 2 --details $R/gpurun_out/$tag/bench.json "$@" > $R/gpurun_out/$tag/log.txt 2>&1
grep -o '"value":[0-9.]*\|"ms_per_step":[0-9.]*' $R/gpurun_out/$tag/log.txt | tr '\n' ' '; echo
f=$(find $R/gpurun_out/$tag -name 't_kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'].split('(')[0][-44:]:44s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e6:7.3f} ms max {float(r['MaxNs'])/1e6:7.3f} pct {r['Percentage']}")
PY
python3 $R/tools/trace_ctx.py $(find $R/gpurun_out/$tag -name 't_kernel_trace.csv' | head -1)
# keep the merge small
find $R/gpurun_out/$tag -name '*.csv' -size +20M -delete
