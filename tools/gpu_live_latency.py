#!/usr/bin/env python3
"""One live 1.6 MS/s stream in pushes of 2^16 ... 2^22 samples: ms per push, stage times, re-run counters (bench.py's live_latency leg with the counters)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wm = importlib.import_module("rtl-wmbus_amd")
cu8 = wm.synth_capture(n_samples=1 << 23, seed=0xC2C2, kinds=7, frames_per_s=20.0)[0]
for n in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 22):
    for extra in ({}, dict(seg_len=16384)):
        with wm.Receiver(n_streams=1, max_push_bytes=2 * n, keep_taps=False, **extra) as rx:
            rows = []
            for k in range(min(10, (1 << 23) // n)):
                t = time.perf_counter(); rx.stage(0, cu8[2 * n * k: 2 * n * (k + 1)]); rx.process(2 * n); rx.collect(); ms = (time.perf_counter() - t) * 1e3
                tm = rx.timing()
                if k: rows.append((ms, tm["clock_ms"], tm["rla_ms"], tm["clock_reruns"], tm["clock_round"], tm["slow_path"]))
        rows.sort()
        m = rows[len(rows) // 2]
        print(n, extra, "median ms %.2f clock %.2f rla %.2f reruns %s rounds %s slow %s | all ms %s" % (m[0], m[1], m[2], m[3], m[4], m[5], [round(r[0], 2) for r in rows]), flush=True)
