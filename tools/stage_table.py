#!/usr/bin/env python3
"""Print the per-context stage times of a bench.py JSON line (stdin or file)."""
import json, sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
keys = ("turn_wait_ms", "demod_ms", "clock_ms", "rla_ms", "gather_ms", "d2h_ms", "gpu_total_ms", "host_decode_ms", "process_wall_ms", "collect_wall_ms")
print("value", d["value"], "ms/step", d["ms_per_step"])
print(" ".join(f"{k[:-3]:>13}" for k in keys))
rows = d.get("stage_ms_mid_step") or d["stage_ms_last_step"]
for t in rows:
    print(" ".join(f"{t.get(k, 0):13.2f}" for k in keys))
print(" ".join(f"{sum(t.get(k, 0) for t in rows) / len(rows):13.2f}" for k in keys), " <- mean")
