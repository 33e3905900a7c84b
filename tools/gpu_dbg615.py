import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib, numpy as np
wm = importlib.import_module("rtl-wmbus_amd")
import oracle_ffi as O
import test_gpu_fuzz as F
from cases import flags_to_kwargs, flags_to_oracle_opts
c = F.make_case(615, 7)
caps = []
for s in range(c["n_streams"]):
    kw = dict(seed=c["seed"] + s, n_samples=c["n"], fs_khz=F.FS[c["d"]], kinds=15, frames_per_s=90.0, amplitude=c["amp"])
    if c["simultaneous"]: kw.update(t1c1_center_khz=325.0, s1_center_khz=-325.0)
    caps.append(wm.synth_capture(**kw)[0])
oo = flags_to_oracle_opts(O, c["flags"])
kw = flags_to_kwargs(c["flags"])
with wm.Receiver(n_streams=c["n_streams"], max_push_bytes=max(c["push"], 4096), **kw, **c["tune"]) as rx:
    texts = rx.run(caps, push_bytes=c["push"])
    print("timing", rx.timing())
    bad = []
    for s in range(c["n_streams"]):
        ref = O.run(caps[s], oo, chips=True)
        for ch in (0, 1):
            for al in (0, 1):
                w, pos = rx.read_chips(ch, al, s)
                oc = F.truncate_runs(ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al)])
                if len(w) != len(oc) or not np.array_equal(pos, oc["sample"]):
                    n = min(len(w), len(oc)); d = np.nonzero(pos[:n] != oc["sample"][:n])[0]
                    bad.append((s, ch, al, len(w), len(oc), int(d[0]) if len(d) else n, int(pos[d[0]]) if len(d) else -1))
    print("bad", bad[:12], len(bad))
    want = O.run_many(caps, oo)
    print("text mismatches", sum(a != b for a, b in zip(texts, want)))
