#!/bin/bash
# kernel trace of the default 4-context bench: steady-state occupancy analysis + one context's timeline
mkdir -p gpurun_out/trace4; export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace4 -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-check > $R/gpurun_out/trace4/log.txt 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $R/gpurun_out/trace4/log.txt | tr '\n' ' '; echo
python3 $R/tools/trace4_analyze.py $R/gpurun_out/trace4/t_kernel_trace.csv
python3 $R/tools/trace_ctx.py $R/gpurun_out/trace4/t_kernel_trace.csv
