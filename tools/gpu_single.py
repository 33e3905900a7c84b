#!/usr/bin/env python3
"""One capture of 2^22 samples pushed whole, several times: wall clock per push and the stages' HIP-event times
(BASELINE configs[1] / configs[2]); run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wm = importlib.import_module("rtl-wmbus_amd")
n = 1 << 22
for name, kw, skw in (("c2", {}, dict(seed=0xC2C2, kinds=7, frames_per_s=20.0)),
                      ("c3", dict(decimation=5, simultaneous=True), dict(seed=0xC3C3, fs_khz=4000, kinds=15, frames_per_s=50.0, t1c1_center_khz=325.0, s1_center_khz=-325.0))):
    cu8 = wm.synth_capture(n_samples=n, **skw)[0]
    for extra in ({}, dict(seg_len=32768, rla_seg_len=8192)):
        with wm.Receiver(n_streams=1, max_push_bytes=2 * n, keep_taps=False, **kw, **extra) as rx:
            rx.push([cu8])
            ms = []
            for _ in range(5):
                t = time.perf_counter(); rx.process(2 * n); rx.collect(); ms.append((time.perf_counter() - t) * 1e3)
            tm = rx.timing()
        print(name, extra, "ms/push", [round(x, 2) for x in ms], {k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items() if k != "chips"}, flush=True)
