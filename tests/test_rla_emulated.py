"""The run-length framer's DEVICE SOURCE (rtl-wmbus_amd/csrc/wm_k2_rla.h) compiled for the host and
run lane by lane (tests/emu/rla_emu.cpp) against the oracle's run-length chips: speculative starts,
hand-off verification and re-runs included.  Needs no GPU; the GPU suite checks the compiled kernel."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from cases import flags_to_oracle_opts

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "emu", "librla_emu.so")
SRC = os.path.join(HERE, "emu", "rla_emu.cpp")
CSRC = os.path.join(ROOT, "rtl-wmbus_amd", "csrc")
F_T1C1, F_S1 = 8, 16                                     # WM_F_* of wm_dev.h


@pytest.fixture(scope="module")
def emu():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("wm_k2_rla.h", "wm_k2_common.h", "wm_dev.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + CSRC, "-Wno-unknown-pragmas", "-o", SO, SRC], check=True)
    L = ctypes.CDLL(SO)
    L.wm_emu_rla.restype = ctypes.c_long
    L.wm_emu_rla.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 7 + [ctypes.c_void_p] * 4
    L.wm_emu_rla_state_bytes.restype = ctypes.c_uint
    L.wm_emu_rla_reset_state.argtypes = [ctypes.c_void_p]
    return L


def run_emulated(emu, bit_rows, pushes, seg_len, lookback, flags=F_T1C1 | F_S1, descending=True, cap=None, expect_overflow=False):
    """bit_rows: [2][M] uint8 slicer bits of one capture; pushes: decimated samples per push."""
    ctypes.c_int.in_dll(emu, "wm_emu_descending").value = int(descending)   # see rla_emu.cpp: launch semantics
    sb = emu.wm_emu_rla_state_bytes()
    carry = np.zeros(2 * sb, np.uint8)
    for r in range(2):
        emu.wm_emu_rla_reset_state(carry[r * sb:].ctypes.data)
    out, m0, reruns = [[], []], 0, 0
    # the product sizes a region as 4 * seg_len + 8 + 8192 chips (wm_api.hip): the bit-length tracker has no floor and
    # 3.1 chips per sample have been seen; beyond that the push fails loudly.  The parity runs here get more room still.
    cap = cap or 8 * seg_len + 8 + 8192
    for M in pushes:
        Mcap = (M + 255) // 256 * 256
        words = np.zeros((2, Mcap // 32), np.uint32)
        for ch in range(2):
            b = np.zeros(Mcap, np.uint8)
            b[:M] = bit_rows[ch][m0:m0 + M]
            words[ch] = np.packbits(b.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).ravel()
        nseg = (M + seg_len - 1) // seg_len
        chips = np.zeros((2, nseg, cap), np.uint32)
        counts = np.zeros((2, nseg), np.uint32)
        err = ctypes.c_uint(0)
        r = emu.wm_emu_rla(words.ctypes.data, 1, M, Mcap, flags, seg_len, lookback, cap, carry.ctypes.data,
                           chips.ctypes.data, counts.ctypes.data, ctypes.byref(err))
        if expect_overflow and err.value == 2:
            return None, -1
        assert r >= 0 and err.value == 0
        reruns += r
        for ch in range(2):
            for s in range(nseg):
                w = chips[ch, s, :counts[ch, s]]
                out[ch].append(np.stack([m0 + s * seg_len + (w >> 3), w & 7], axis=1))
        m0 += M
    return [np.concatenate(o) if o else np.zeros((0, 2), np.uint32) for o in out], reruns


def oracle_rla_chips(ref, ch):
    oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == 0)]
    return np.stack([oc["sample"].astype(np.uint32), oc["value"].astype(np.uint32)], axis=1)


@pytest.mark.parametrize("seg_len,lookback", [(8192, 1024), (1024, 64), (2048, 32)])
def test_device_source_on_host_matches_oracle_bundled_capture(emu, oracle, samples, seg_len, lookback):
    cu8 = samples["samples2"]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    got, reruns = run_emulated(emu, ref["bit"], [ref["m"]], seg_len, lookback)
    for ch in (0, 1):
        assert np.array_equal(got[ch], oracle_rla_chips(ref, ch)), ch
    if seg_len < 8192:
        assert reruns > 0                                    # the re-run path really ran


def test_device_source_on_host_matches_oracle_across_pushes_and_synthetic(emu, oracle, wm):
    rng = np.random.default_rng(5 + int(os.environ.get("WMBUS_EMU_SEED", "0")))
    for k in range(int(os.environ.get("WMBUS_EMU_N", "8"))):          # more for a bug hunt
        cu8 = wm.synth_capture(seed=int(rng.integers(1, 1 << 30)), n_samples=1 << 18, kinds=int(rng.choice([15, 15, 8, 7])), frames_per_s=120.0,
                               amplitude=float(rng.choice([8.0, 25.0, 60.0])), noise_sigma=float(rng.choice([0.5, 3.0, 3.0, 10.0])))[0]
        if k % 2:                                            # a stretch of exact silence (long runs, resets)
            a = int(rng.integers(0, cu8.size // 2)) & ~1
            cu8[a:a + int(rng.integers(4096, cu8.size // 3))] = int(rng.choice([127, 128]))
        if k % 7 == 6:                                       # silence longer than WM_RLA_RUN_LIMIT chips, then signal again
            cu8[cu8.size // 16 & ~1: cu8.size * 15 // 16 & ~1] = 128
        if k % 5 == 4:                                       # a slow square wave: chips far off the nominal rate
            a = int(rng.integers(0, cu8.size // 2)) & ~1
            per = int(rng.integers(3, 200))
            n_sq = min(cu8.size - a, 60000) // 2
            ph = (np.arange(n_sq) // per) % 2
            cu8[a:a + 2 * n_sq:2] = 128 + 60 * np.cos(2 * np.pi * 0.03 * np.arange(n_sq) * (2 * ph - 1))
            cu8[a + 1:a + 2 * n_sq:2] = 128 + 60 * np.sin(2 * np.pi * 0.03 * np.arange(n_sq) * (2 * ph - 1))
        ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
        M = ref["m"]
        # pushes are multiples of 4096 B = 2048 IQ samples: whole kilo-samples at d = 2, ragged counts at d = 3, 5, ...
        q = 2048 if k % 3 else 1
        cuts = sorted(set(int(x) // q * q for x in rng.integers(2048, M, 3)))
        pushes = [b - a for a, b in zip([0] + cuts, cuts + [M]) if b > a]
        got, _ = run_emulated(emu, ref["bit"], pushes, int(rng.choice([1024, 4096, 8192])), int(rng.choice([32, 256, 1024])), descending=bool(k % 5))
        for ch in (0, 1):
            want = oracle_rla_chips(ref, ch)
            # the kernel materialises at most 8192 chips per edge (tests/test_gpu_fuzz.py: truncate_runs)
            new_edge = np.concatenate([[True], want[1:, 0] != want[:-1, 0]]) if len(want) else np.zeros(0, bool)
            start = np.maximum.accumulate(np.where(new_edge, np.arange(len(want)), 0)) if len(want) else np.zeros(0, int)
            want = want[np.arange(len(want)) - start < 8192]
            assert np.array_equal(got[ch], want), (k, ch)


def test_chip_region_overflow_is_reported_not_silent(emu, oracle, wm):
    """A strong square-wave FM interferer (period 29 samples) drags the T1/C1 bit-length tracker far below one sample
    per chip: the reference emits three chips per sample for a while.  With a region of one chip per sample (the product's size until
    this case was found; it is four chips per sample now) the kernel must raise the overflow flag (wmbus_process then fails with WMBUS_EOVERFLOW); with enough room it is exact."""
    # found by a long emulation campaign (WMBUS_EMU_SEED=1030, case 904): quiet capture (noise 0.5 LSB), then the interferer
    cu8 = wm.synth_capture(seed=912168056, n_samples=1 << 18, kinds=15, frames_per_s=120.0, amplitude=60.0, noise_sigma=0.5)[0]
    a, per = 147918, 29
    n_sq = min(cu8.size - a, 60000) // 2
    t = np.arange(n_sq); ph = (t // per) % 2
    cu8[a:a + 2 * n_sq:2] = 128 + 60 * np.cos(2 * np.pi * 0.03 * t * (2 * ph - 1))
    cu8[a + 1:a + 2 * n_sq:2] = 128 + 60 * np.sin(2 * np.pi * 0.03 * t * (2 * ph - 1))
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    oc = oracle_rla_chips(ref, 0)
    assert np.bincount(oc[:, 0] // 8192).max() > 8192 + 8 + 8192          # the oracle really emits more than a region holds
    got, reruns = run_emulated(emu, ref["bit"], [ref["m"]], 8192, 1024, cap=8192 + 8 + 8192, expect_overflow=True)
    assert got is None and reruns == -1                                        # flagged
    got, _ = run_emulated(emu, ref["bit"], [ref["m"]], 8192, 1024)
    want = oc
    new_edge = np.concatenate([[True], want[1:, 0] != want[:-1, 0]])
    start = np.maximum.accumulate(np.where(new_edge, np.arange(len(want)), 0))
    assert np.array_equal(got[0], want[np.arange(len(want)) - start < 8192])
