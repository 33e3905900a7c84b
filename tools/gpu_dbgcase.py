"""Re-run one configuration of tests/test_gpu_fuzz.py and say where it differs: tools/gpu_dbgcase.py K [SEED_OFFSET]"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib, numpy as np
wm = importlib.import_module("rtl-wmbus_amd")
import oracle_ffi as O
import test_gpu_fuzz as F
from cases import flags_to_kwargs, flags_to_oracle_opts
c = F.make_case(int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else None)
print(c, flush=True)
rng = np.random.default_rng(c["seed"])
caps = []
for s in range(c["n_streams"]):
    kw = dict(seed=c["seed"] + s, n_samples=c["n"], fs_khz=F.FS[c["d"]], kinds=15, frames_per_s=90.0, amplitude=c["amp"])
    if c["simultaneous"]: kw.update(t1c1_center_khz=325.0, s1_center_khz=-325.0)
    cu8 = wm.synth_capture(**kw)[0]
    if c["silence"]:
        a = int(rng.integers(0, cu8.size // 2)) & ~1
        cu8[a:a + int(rng.integers(4096, cu8.size // 3))] = int(rng.choice([127, 128]))
    caps.append(cu8)
oo = flags_to_oracle_opts(O, c["flags"]); oo.prefilter = c["prefilter"]
kw = flags_to_kwargs(c["flags"])
want = O.run_many(caps, oo)
with wm.Receiver(n_streams=c["n_streams"], max_push_bytes=max(c["push"], 4096), prefilter=c["prefilter"], **kw, **c["tune"]) as rx:
    total = caps[0].size // 4096 * 4096
    texts = [""] * c["n_streams"]
    for off in range(0, total, c["push"]):
        n = min(c["push"], total - off)
        rx.push([a[off:off + n] for a in caps])
        print("push", off, rx.timing(), flush=True)
        for ln in rx.lines():
            texts[ln["stream"]] += ln["text"]
    print("text mismatches", [s for s in range(c["n_streams"]) if texts[s] != want[s]][:8])
