#!/usr/bin/env python3
"""Run one randomised configuration (tests/test_gpu_fuzz.py) and show where the chip streams differ.
usage: WMBUS_FUZZ_SEED=s tools/fuzz_debug.py k"""
import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
k = int(sys.argv[1])
os.environ["WMBUS_FUZZ_N"] = str(k + 1)
import test_gpu_fuzz as T
import oracle_ffi as O
from cases import flags_to_kwargs, flags_to_oracle_opts
wm = importlib.import_module("rtl-wmbus_amd")
c = T.CASES[k]; print(c)
rng = np.random.default_rng(c["seed"])
caps = []
for s in range(c["n_streams"]):
    kw = dict(seed=c["seed"] + s, n_samples=c["n"], fs_khz=T.FS[c["d"]], kinds=15, frames_per_s=90.0, amplitude=c["amp"])
    if c["simultaneous"]: kw.update(t1c1_center_khz=325.0, s1_center_khz=-325.0)
    cu8 = wm.synth_capture(**kw)[0]
    if c["silence"]:
        a = int(rng.integers(0, cu8.size // 2)) & ~1
        cu8[a:a + int(rng.integers(4096, cu8.size // 3))] = int(rng.choice([127, 128]))
    caps.append(cu8)
oo = flags_to_oracle_opts(O, c["flags"]); oo.prefilter = c["prefilter"]
kw = flags_to_kwargs(c["flags"])
with wm.Receiver(n_streams=c["n_streams"], max_push_bytes=max(c["push"], 4096), prefilter=c["prefilter"], **kw, **c["tune"]) as rx:
    texts = rx.run(caps, push_bytes=c["push"])
    print(rx.timing())
    s = c["n_streams"] - 1
    ref = O.run(caps[s], oo, taps=True, chips=True)
    total = caps[s].size // 4096 * 4096
    last = (total - 1) // c["push"] * c["push"] if c["push"] < total else 0
    m_first = (last // 2) // c["d"]
    for ch in (0, 1):
        for al in (0, 1):
            try:
                w, pos = rx.read_chips(ch, al, s)
            except Exception as e:
                print(ch, al, e); continue
            oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al) & (ref["chips"]["sample"] >= m_first)]
            oc = T.truncate_runs(oc)
            n = min(len(w), len(oc))
            bad = np.nonzero((pos[:n] != oc["sample"][:n]) | ((w[:n] & 0xFF) != oc["value"][:n]))[0]
            print(f"chain {ch} algo {al}: hip {len(w)} oracle {len(oc)} first mismatch", bad[:1])
            if len(bad):
                i = int(bad[0]); seg = c["tune"].get("seg_len", 32768) if al else c["tune"].get("rla_seg_len", 8192)
                print("   around:", "hip", list(zip(pos[i-2:i+4].tolist(), (w[i-2:i+4] & 0xFF).tolist())), "oracle", list(zip(oc["sample"][i-2:i+4].tolist(), oc["value"][i-2:i+4].tolist())))
                print("   segment index", (int(oc['sample'][i]) - m_first) // seg, "offset in segment", (int(oc['sample'][i]) - m_first) % seg)
    print("text equal:", texts[s] == ref["text"])
