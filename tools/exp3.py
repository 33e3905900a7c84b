import importlib, sys, time, os, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
wm = importlib.import_module("rtl-wmbus_amd")
S, n = int(sys.argv[1]), int(sys.argv[2])
caps = [wm.synth_capture(seed=0xC0FFEE + s, n_samples=n, kinds=7, frames_per_s=20.0)[0] for s in range(min(S, 64))]
def run(label, **kw):
    env = kw.pop("env", {})
    for k, v in env.items(): os.environ[k] = v
    rx = wm.Receiver(n_streams=S, max_push_bytes=2 * n, **kw)
    for s in range(S): rx.stage(s, caps[s % len(caps)])
    best = None
    for it in range(3):
        t = time.perf_counter(); rx.process(2 * n); rx.collect(); dt = time.perf_counter() - t
        tm = rx.timing(); tm["wall_ms"] = dt * 1e3
        if best is None or dt * 1e3 < best["wall_ms"]: best = tm
    rx.close()
    for k in env: os.environ.pop(k)
    print(label, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in best.items()}, "Gs/s", round(S * n / best["wall_ms"] / 1e6, 1), flush=True)
for spec in sys.argv[3:]:
    kw = {}
    for kv in spec.split(","):
        if not kv: continue
        k, v = kv.split("=")
        if k.startswith("WMBUS"): kw.setdefault("env", {})[k] = v
        else: kw[k] = int(v)
    run(spec or "default", **kw)
