import importlib, sys, time, os, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
wm = importlib.import_module("rtl-wmbus_amd")
import oracle_ffi as O
from cases import *
import json
BUNDLED = json.load(open("tests/golden/bundled.json"))
name, flags = S2, ["-d", "3", "-v"]
cu8 = np.fromfile("tests/golden/samples/" + name, np.uint8)
ref = O.run(cu8, flags_to_oracle_opts(O, flags), taps=True, chips=True)
for kw in (dict(), dict(rla_seg_len=65536), dict(seg_len=65536), dict(seg_len=65536, rla_seg_len=65536)):
    with wm.Receiver(n_streams=1, max_push_bytes=4 << 20, **flags_to_kwargs(flags), **kw) as rx:
        out = rx.run(cu8)[0]
        print(kw, out == BUNDLED[f"{name}|{' '.join(flags)}"], rx.timing())
        print(out)
        for ch in (0, 1):
            for al in (0, 1):
                w, pos = rx.read_chips(ch, al, 0)
                oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al)]
                n = min(len(w), len(oc))
                bad = np.nonzero(((w[:n] & 0xFF) != oc["value"][:n]) | (pos[:n] != oc["sample"][:n]) | (((w[:n] >> 8) & 0xFF) != oc["rssi"][:n]))[0]
                print("  chain", ch, "algo", al, len(w), len(oc), "first bad", bad[:3], (pos[bad[:3]], oc["sample"][bad[:3]]) if len(bad) else "")
print(BUNDLED[f"{name}|{' '.join(flags)}"])
