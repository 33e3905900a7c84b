#!/usr/bin/env python3
"""What a lock-step wave pays in the run-length framer: edge trips per 64 samples for the unluckiest of 64 lanes, by step length.
Runs the device source on the host (tests/emu/rla_emu.cpp, which counts a lane's trips per 64 samples) over 64 captures of the
bench workload -- the 64 lanes of one wave -- and takes, per step of 64 / 128 / 256 / 512 samples, the largest count among them.
No GPU.  DESIGN_HISTORY.md section 4 quotes the output."""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
wm = importlib.import_module("rtl-wmbus_amd")
import oracle_ffi
from cases import flags_to_oracle_opts
CSRC = os.path.join(ROOT, "rtl-wmbus_amd", "csrc"); SO = os.path.join(ROOT, "tests", "emu", "librla_emu.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + CSRC, "-Wno-unknown-pragmas", "-o", SO, os.path.join(ROOT, "tests", "emu", "rla_emu.cpp")], check=True)
emu = ctypes.CDLL(SO)
emu.wm_emu_rla.restype = ctypes.c_long
emu.wm_emu_rla.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 7 + [ctypes.c_void_p] * 4
emu.wm_emu_rla_set_spill.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
oracle_ffi.lib(); oracle = oracle_ffi
N, LANES = 1 << 18, 64
tabs = []
for i in range(LANES):
    cu8 = wm.synth_capture(seed=0xC0FFEE + i, n_samples=N, kinds=wm.T1 | wm.C1A | wm.C1B, frames_per_s=20.0)[0]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=False)
    M = ref["m"]; Mcap = (M + 255) // 256 * 256
    words = np.zeros((2, Mcap // 32), np.uint32)
    for ch in range(2):
        b = np.zeros(Mcap, np.uint8); b[:M] = ref["bit"][ch]
        words[ch] = np.packbits(b.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).ravel()
    sb = emu.wm_emu_rla_state_bytes(); carry = np.zeros(2 * sb, np.uint8)
    for r in range(2): emu.wm_emu_rla_reset_state(ctypes.c_void_p(carry[r * sb:].ctypes.data))
    cap = 8 * M + 8 + 8192
    chips = np.zeros((2, 1, cap), np.uint32); counts = np.zeros((2, 1), np.uint32); err = ctypes.c_uint(0)
    tab = np.zeros((2, Mcap // 64), np.uint32)
    ctypes.c_void_p.in_dll(emu, "wm_emu_trip_tab").value = tab.ctypes.data
    ctypes.c_uint.in_dll(emu, "wm_emu_trip_words").value = Mcap // 64
    emu.wm_emu_rla_set_spill(None, 0, None, None, None)
    r = emu.wm_emu_rla(words.ctypes.data, 1, M, Mcap, 8 | 16, M, 1024, cap, carry.ctypes.data, chips.ctypes.data, counts.ctypes.data, ctypes.byref(err))
    assert r == 0 and err.value == 0
    tabs.append(tab)
t = np.stack(tabs).astype(np.int64)                        # [lane][chain][64-sample word]
for ch, name in ((0, "T1/C1 chain"), (1, "S1 chain")):
    x = t[:, ch, :t.shape[2] // 8 * 8]
    print("%s: %.2f edges per 64 samples and lane on average" % (name, x.mean()))
    for step in (64, 128, 256, 512):
        k = step // 64
        per_step = x.reshape(LANES, -1, k).sum(axis=2)       # a lane's trips per step
        print("  step %3d samples: the wave pays %.2f trips per 64 samples (largest of %d lanes per step)" % (step, per_step.max(axis=0).mean() / k, LANES))
