#!/bin/bash
# The PRODUCT's rate: wall-clock `rtl_wmbus_hip FILE...` (batch mode) over 1024 capture files in /dev/shm, beside
# `bench.py --from-host` (the same host-sourced path driven from Python) and the PCIe bound.  Run on the GPU box:
#     gpurun -- ./tools/bench_cli.sh            -> gpurun_out/cli_rate.json (copy to profiles/cli_rate.json)
# 32 distinct synthetic captures of PASSES x 8 MiB each; the 1024 file names are symlinks onto them (the CLI opens every
# name on its own; the page cache serves the bytes), so /dev/shm holds 32 x PASSES x 8 MiB, not 1024 x.
set -e
cd "$(dirname "$0")/.."
PASSES=${PASSES:-8}; NFILES=${NFILES:-1024}; DISTINCT=${DISTINCT:-32}
D=/dev/shm/wmbus_cli_bench; rm -rf $D; mkdir -p $D gpurun_out
python3 - <<PY
import importlib, os, sys
sys.path.insert(0, os.getcwd())
wm = importlib.import_module("rtl-wmbus_amd")
n = $PASSES * (1 << 22)
for i in range($DISTINCT):
    wm.synth_capture(seed=0xC0FFEE + i, n_samples=n, kinds=wm.T1 | wm.C1A | wm.C1B, frames_per_s=20.0)[0].tofile("$D/src%02d.cu8" % i)
for i in range($NFILES):
    os.symlink("$D/src%02d.cu8" % (i % $DISTINCT), "$D/f%04d.cu8" % i)
PY
FILES=$(ls $D/f*.cu8)
run() {  # $1 = tag, rest = CLI arguments
  tag=$1; shift
  s=$(date +%s.%N)
  ./rtl-wmbus_amd/rtl_wmbus_hip -S "$@" $FILES > $D/out_$tag.txt 2> $D/err_$tag.txt
  e=$(date +%s.%N)
  echo "$tag wall $(python3 -c "print(round($e - $s, 3))") s, lines $(wc -l < $D/out_$tag.txt)"; grep "total:" $D/err_$tag.txt
}
run warm -v $CLI_ARGS         # first touch of the page cache, HIP start-up
run a -v $CLI_ARGS
run b -v $CLI_ARGS
python3 - <<PY
import json, re, subprocess, os
def parse(tag):
    t = open("$D/err_%s.txt" % tag).read()
    m = re.search(r"total: (\d+) files on (\d+) device\(s\), (\d+) samples; decode ([\d.]+) s = ([\d.]+) Msamples/s; with set-up .*? ([\d.]+) s = ([\d.]+) Msamples/s", t)
    return dict(files=int(m.group(1)), samples=int(m.group(3)), decode_s=float(m.group(4)), decode_msamples_s=float(m.group(5)),
                with_setup_s=float(m.group(6)), with_setup_msamples_s=float(m.group(7)), lines=sum(1 for _ in open("$D/out_%s.txt" % tag)))
runs = [parse("a"), parse("b")]
best = max(runs, key=lambda r: r["decode_msamples_s"])
out = {"collected_at": os.environ.get("WMBUS_COMMIT", "unknown"), "command": "rtl_wmbus_hip -v -S f0000.cu8 ... f%04d.cu8 (batch mode, default 2 MiB pushes, $NFILES files of $PASSES x 8 MiB in /dev/shm)" % ($NFILES - 1),
       "value": best["decode_msamples_s"], "unit": "Msamples/s", "runs": runs,
       "pcie_bound_msamples_s": 24800.0, "note": "decode = wmbus_batch_run wall clock (file reads into page-locked slabs, H2D, kernels, host decode, printing); "
       "with_setup adds opening the contexts and allocating the page-locked staging"}
fh = subprocess.run("true" if os.environ.get("CLI_SKIP_FROM_HOST") else "timeout 600 python bench.py --from-host --steps 8 --warmup 1 --no-cpu-baseline --no-check", shell=True, capture_output=True, text=True)
try:
    out["bench_from_host_msamples_s"] = json.loads(fh.stdout.strip().splitlines()[-1])["value"]
    out["cli_over_bench_from_host"] = round(out["value"] / out["bench_from_host_msamples_s"], 3)
except Exception as e:
    out["bench_from_host_error"] = repr(e) + fh.stderr[-300:]
json.dump(out, open("gpurun_out/cli_rate.json", "w"), indent=1)
print(json.dumps(out))
PY
[ -n "$KEEP_FILES" ] || rm -rf $D
