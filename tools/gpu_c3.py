#!/usr/bin/env python3
"""configs[2] at batch size: S captures at 4.0 MS/s through -d 5 -s; rate, per-stage means, re-run counters (bench.py's c3_batch leg on its own)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import oracle_ffi as O
wm = importlib.import_module("rtl-wmbus_amd")
shard = importlib.import_module("rtl-wmbus_amd.shard")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tune = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}      # e.g. rla_seg_len=32768 rla_lookback=4096
r = bench.leg_c3_batch(wm, O, shard, S, 1 << 22, 0, 5, **tune)
print(json.dumps({"tune": tune, "value": r["value"], "ms_per_step": r["ms_per_step"], "parity": r["parity"], "stages": r["stage_ms_mean_per_context_push"]}))
