import importlib, os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wm = importlib.import_module("rtl-wmbus_amd")
n = 1 << 22
cu8 = wm.synth_capture(n_samples=n, seed=0xC2C2, kinds=7, frames_per_s=20.0)[0]
for cw in (1, 4):
    for extra in ({}, dict(seg_len=16384), dict(seg_len=32768), dict(seg_len=65536)):
        with wm.Receiver(n_streams=1, max_push_bytes=2 * n, keep_taps=False, clock_waves=cw, **extra) as rx:
            rx.push([cu8])
            ms = []
            for _ in range(5):
                t = time.perf_counter(); rx.process(2 * n); rx.collect(); ms.append((time.perf_counter() - t) * 1e3)
            tm = rx.timing()
        print("clock_waves", cw, extra, "ms/push", [round(x, 2) for x in ms], {k: round(v, 3) if isinstance(v, float) else v for k, v in tm.items() if k in ("clock_ms", "rla_ms", "gather_ms", "demod_ms", "gpu_total_ms", "clock_reruns", "slow_path")}, flush=True)
