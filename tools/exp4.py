import importlib, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
wm = importlib.import_module("rtl-wmbus_amd")
import oracle_ffi as O
cu8, frames = wm.synth_capture(seed=0xC0FFEE, n_samples=1 << 19, kinds=15, frames_per_s=60.0)
ref = O.run(cu8, O.make_opts(), taps=True)
m = ref["m"]
for kw in (dict(seg_len=16384, warmup_t1c1=8192, warmup_s1=8192), dict()):
    with wm.Receiver(n_streams=1, max_push_bytes=1 << 20, **kw) as rx:
        text = rx.run(cu8)[0]
        print(kw, "text", text == ref["text"], "M", m)
        for ch in (0, 1):
            r = rx.read_tap("rssi", ch, 0, m); want = ref["rssi"][ch].astype(np.uint32).astype(np.uint8)
            bad = np.nonzero(r != want)[0]
            print(" ch", ch, "n bad", len(bad), bad[:10], r[bad[:10]], want[bad[:10]], len(r), len(want))
