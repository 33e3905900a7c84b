#!/bin/bash
# instruction-cache counters of a quick bench run, per kernel:  tools/gpu_icache.sh TAG [bench flags]
tag=$1; shift
mkdir -p gpurun_out/$tag; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/$tag -o c --output-format csv -- python $R/bench.py --quick --steps 4 --warmup 1 "$@" > $R/gpurun_out/$tag/log.txt 2>&1
f=$(find $R/gpurun_out/$tag -name 'c_counter_collection.csv' | head -1)
python3 - $f <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[k] += 1
print("%-42s %12s %12s %8s %12s" % ("kernel", "icache req", "misses", "miss %", "ifetch"))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQC_ICACHE_REQ', 0))[:14]:
    req, mis = v.get('SQC_ICACHE_REQ', 0), v.get('SQC_ICACHE_MISSES', 0)
    print("%-42s %12.3g %12.3g %8.2f %12.3g" % (k, req, mis, 100 * mis / max(req, 1), v.get('SQ_IFETCH', 0)))
PY
find $R/gpurun_out/$tag -name '*.csv' -size +20M -delete
