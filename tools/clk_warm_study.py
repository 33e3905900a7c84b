#!/usr/bin/env python3
"""Host-only study behind the clock kernel's warm-up scheme (tools/clk_warm_study.cpp): certification rate of the
speculative segment starts and the length of the re-runs, on the soft symbols of the bench workload's captures
(oracle taps; TEST INFRASTRUCTURE use of the oracle, nothing of this is in the product path).

    python tools/clk_warm_study.py [n_captures] > profiles/r05_clock_warmup_study.txt
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
import oracle_ffi as O

wm = importlib.import_module("rtl-wmbus_amd")
SO = os.path.join(ROOT, "tools", "libclk_warm_study.so")
SRC = os.path.join(ROOT, "tools", "clk_warm_study.cpp")
if not os.path.exists(SO) or os.path.getmtime(SRC) > os.path.getmtime(SO):
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-o", SO, SRC], check=True)
L = ctypes.CDLL(SO)
L.clk_warm_study.restype = ctypes.c_int
L.clk_warm_study.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int] + [ctypes.c_uint] * 4 + [ctypes.c_void_p, ctypes.c_uint]


def main():
    ncap = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n = 1 << 22
    seg, ck, nl = 32768, 2048, 18
    rows = []
    for s in range(ncap):
        cu8 = wm.synth_capture(seed=0xC0FFEE + s, n_samples=n, kinds=wm.T1 | wm.C1A | wm.C1B, frames_per_s=20.0)[0]
        ref = O.run(cu8, O.make_opts(), taps=True)
        rows.append([np.ascontiguousarray(ref["dphi_fir"][ch][:ref["m"]], np.float32) for ch in (0, 1)])
    print(f"# {ncap} captures of the bench workload (seed 0xC0FFEE + s), 2^22 samples, segments of {seg}, checkpoints every {ck}")
    print("# leave[i]: hand-offs certified at the boundary (i = 0) / re-run meets the speculative pass at checkpoint i / not within the segment (last)")
    schemes = {0: [(12288, 12288), (12288, 8192), (12288, 6144), (12288, 4096), (12288, 2048), (16384, 4096), (16384, 8192), (8192, 8192), (16384, 16384)],
               1: [(24576, 24576), (24576, 16384), (24576, 12288), (24576, 8192), (24576, 4096), (32768, 8192), (32768, 16384), (28672, 12288), (32768, 32768), (16384, 16384)]}
    stag = {0: [(12288, 2048, 2048), (12288, 4096, 2048), (12288, 4096, 4096), (12288, 2048, 6144), (16384, 4096, 4096)],
            1: [(24576, 4096, 4096), (24576, 8192, 4096), (24576, 8192, 8192), (24576, 4096, 12288), (32768, 8192, 8192)]}
    if "--stagger" in sys.argv:
        print("# STAGGERED warm-ups: section 1 starts A samples, section 2 A + B samples behind section 0 (all exact); ops: 5 / 13 / 20 per sample in the three phases, 27 in the segment")
        for ch in (0, 1):
            for W, A, B in stag[ch]:
                leave = np.zeros(nl, np.uint32)
                tot = 0
                for r in rows:
                    tot += L.clk_warm_study(r[ch].ctypes.data, r[ch].size, ch, seg, W, (1 << 30) | (A << 15) | B, ck, leave.ctypes.data, nl)
                cost = (A * 5 + B * 13 + (W - A - B) * 20 + seg * 27) / (W * 20.0 + seg * 27.0)
                print(f"chain {ch} W {W:6d} section 1 at +{A:5d}, section 2 at +{A + B:5d}: boundaries {tot:5d} failed {100.0 * (tot - leave[0]) / tot:6.2f} %  "
                      f"first-pass instructions x{cost:5.3f}  leave {' '.join(str(v) for v in leave)}", flush=True)
        return
    for ch in (0, 1):
        for W, E in schemes[ch]:
            leave = np.zeros(nl, np.uint32)
            tot = 0
            for r in rows:
                tot += L.clk_warm_study(r[ch].ctypes.data, r[ch].size, ch, seg, W, E, ck, leave.ctypes.data, nl)
            cost = ((W - E) * 13 + E * 31 + seg * 31) / ((W + seg) * 31.0) if E < W else 1.0
            cost_abs = ((W - min(E, W)) * 13 + min(E, W) * 31 + seg * 31) / (seg * 31.0)
            print(f"chain {ch} W {W:6d} exact tail {min(E, W):6d}: boundaries {tot:5d} failed {100.0 * (tot - leave[0]) / tot:6.2f} %  "
                  f"first-pass ops per segment sample {cost_abs:5.3f} (x{cost:5.3f})  leave {' '.join(str(v) for v in leave)}", flush=True)


if __name__ == "__main__":
    main()
