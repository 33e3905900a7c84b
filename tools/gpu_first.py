import importlib, sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
wm = importlib.import_module("rtl-wmbus_amd")
import oracle_ffi as O
print("devices", wm.device_count())
cu8 = np.fromfile("tests/golden/samples/rtlsdr_868.950M_1M6_samples2.cu8", np.uint8)
rx = wm.Receiver(n_streams=1, max_push_bytes=4 << 20)
t = time.time(); out = rx.run(cu8)[0]; print("run s", time.time() - t)
print(out)
ref = O.run(cu8, O.make_opts(), taps=True, chips=True)
print("MATCH" if out == ref["text"] else "MISMATCH"); print(ref["text"])
print(rx.timing())
M = ref["m"]
for ch in (0, 1):
    d = rx.read_tap("dphi", ch, 0, M); r = rx.read_tap("rssi", ch, 0, M); b = rx.read_tap("bits", ch, 0, M)
    print("chain", ch, "dphi maxabs diff", np.abs(d - ref["dphi"][ch]).max(), "bitexact", np.array_equal(d.view(np.uint32), ref["dphi"][ch].view(np.uint32)),
          "rssi eq", np.array_equal(r, ref["rssi"][ch].astype(np.uint32).astype(np.uint8)), "bits eq", np.array_equal(b, ref["bit"][ch]))
    for al in (0, 1):
        w, pos = rx.read_chips(ch, al, 0)
        oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al)]
        ok = len(w) == len(oc) and np.array_equal(w & 0xFF, oc["value"]) and np.array_equal((w >> 8) & 0xFF, oc["rssi"]) and np.array_equal(pos, oc["sample"])
        print("  algo", al, "chips gpu", len(w), "oracle", len(oc), "equal", ok)
