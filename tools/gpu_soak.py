"""Soak: a long capture through the CLI in many small pushes (carried state flips thousands of times, decoders persist
across pushes, every burst straddles push boundaries sooner or later) against the oracle; then 100 bench steps."""
import sys, os, subprocess, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib, numpy as np
wm = importlib.import_module("rtl-wmbus_amd")
import oracle_ffi as O
n = 1 << 25                                             # 32 Mi samples = 64 MiB = 21 s of air time at 1.6 MS/s
cu8 = wm.synth_capture(seed=777, n_samples=n, kinds=15, frames_per_s=60.0, amplitude=40.0, max_frames=4096)[0]
t = time.time(); want = O.run(cu8, O.make_opts())["text"]; print("oracle", round(time.time() - t, 1), "s", len(want.splitlines()), "lines", flush=True)
env = dict(os.environ, WMBUS_FIXED_TS="1")
for b in (65536, 4096 * 5, 1 << 20):
    t = time.time()
    p = subprocess.run([wm.CLI_PATH, "-v", "-B", str(b)], input=cu8.tobytes(), capture_output=True, env=env)
    print("-B", b, "rc", p.returncode, "identical", p.stdout.decode() == want, round(time.time() - t, 1), "s", p.stderr.decode()[-200:], flush=True)
