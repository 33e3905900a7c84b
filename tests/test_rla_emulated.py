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
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + CSRC, "-Wno-unknown-pragmas"] + os.environ.get("WMBUS_EMU_CFLAGS", "").split() +
                       ["-o", SO, SRC], check=True)
    L = ctypes.CDLL(SO)
    L.wm_emu_rla.restype = ctypes.c_long
    L.wm_emu_rla.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 7 + [ctypes.c_void_p] * 4
    L.wm_emu_rla_set_spill.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.wm_emu_rla_state_bytes.restype = ctypes.c_uint
    L.wm_emu_rla_reset_state.argtypes = [ctypes.c_void_p]
    return L


def run_emulated(emu, bit_rows, pushes, seg_len, lookback, flags=F_T1C1 | F_S1, descending=True, cap=None, spill_words=None, chains=True):
    """bit_rows: [2][M] uint8 slicer bits of one capture; pushes: decimated samples per push."""
    ctypes.c_int.in_dll(emu, "wm_emu_descending").value = int(descending)   # see rla_emu.cpp: launch semantics
    ctypes.c_int.in_dll(emu, "wm_emu_chains").value = int(chains)           # a listed lane walks its chain of listed segments (K2Args.bad)
    sb = emu.wm_emu_rla_state_bytes()
    carry = np.zeros(2 * sb, np.uint8)
    for r in range(2):
        emu.wm_emu_rla_reset_state(carry[r * sb:].ctypes.data)
    out, m0, reruns = [[], []], 0, 0
    # spill_words None: one roomy region per segment (no spill storage); else the product's layout: a primary region of
    # `cap` chips (wm_api.hip: seg_len / 2 + 8) and the chunk arena of WmSpill behind it.  Returns chips, re-runs and,
    # with spill storage, the device's warning word of the last push.
    cap = cap or 8 * seg_len + 8 + 8192
    CH, LV = emu.wm_emu_spill_chunk(), emu.wm_emu_spill_levels()
    warn = 0
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
        if spill_words is not None:
            arena = np.zeros(max(1, spill_words), np.uint32); chain = np.zeros((2, nseg, LV), np.uint32); nchain = np.zeros(2 * nseg + 1, np.uint32)
            emu.wm_emu_rla_set_spill(arena.ctypes.data, spill_words, chain.ctypes.data, nchain.ctypes.data, nchain[2 * nseg:].ctypes.data)
        else:
            emu.wm_emu_rla_set_spill(None, 0, None, None, None)
        r = emu.wm_emu_rla(words.ctypes.data, 1, M, Mcap, flags, seg_len, lookback, cap, carry.ctypes.data,
                           chips.ctypes.data, counts.ctypes.data, ctypes.byref(err))
        assert r >= 0 and err.value in ((0, 8) if spill_words is not None else (0,))      # 8 = WM_ERR_CHIP_TRUNC: chips dropped, a warning
        warn |= err.value
        reruns += r
        for ch in range(2):
            for s in range(nseg):
                n = int(counts[ch, s])
                w = chips[ch, s, :min(n, cap)]
                if n > cap:                                  # the segment's spill chain
                    rest = n - cap
                    assert nchain[ch * nseg + s] * CH >= rest
                    w = np.concatenate([w] + [arena[chain[ch, s, l]: chain[ch, s, l] + min(CH, rest - l * CH)] for l in range((rest + CH - 1) // CH)])
                out[ch].append(np.stack([m0 + s * seg_len + (w >> 3), w & 7], axis=1))
        m0 += M
    res = [np.concatenate(o) if o else np.zeros((0, 2), np.uint32) for o in out]
    return (res, reruns) if spill_words is None else (res, reruns, warn)


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


def truncate_runs(want):
    """the kernel materialises at most 8192 chips per edge"""
    new_edge = np.concatenate([[True], want[1:, 0] != want[:-1, 0]]) if len(want) else np.zeros(0, bool)
    start = np.maximum.accumulate(np.where(new_edge, np.arange(len(want)), 0)) if len(want) else np.zeros(0, int)
    return want[np.arange(len(want)) - start < 8192]


def interferer_capture(wm):
    """A strong square-wave FM interferer (period 29 samples) drags the T1/C1 bit-length tracker far below one sample per
    chip: the reference emits three chips per sample for a while.  Found by a long emulation campaign (WMBUS_EMU_SEED=1030,
    case 904): quiet capture (noise 0.5 LSB), then the interferer."""
    cu8 = wm.synth_capture(seed=912168056, n_samples=1 << 18, kinds=15, frames_per_s=120.0, amplitude=60.0, noise_sigma=0.5)[0]
    a, per = 147918, 29
    n_sq = min(cu8.size - a, 60000) // 2
    t = np.arange(n_sq); ph = (t // per) % 2
    cu8[a:a + 2 * n_sq:2] = 128 + 60 * np.cos(2 * np.pi * 0.03 * t * (2 * ph - 1))
    cu8[a + 1:a + 2 * n_sq:2] = 128 + 60 * np.sin(2 * np.pi * 0.03 * t * (2 * ph - 1))
    return cu8


def test_a_burst_longer_than_a_segment_is_settled_by_one_chain_walk(emu, oracle, wm):
    """S1 telegrams are 30-100 ms long, a run-length segment 10 ms (and the tests' short segments 1.3 ms): every segment inside
    a burst starts wrongly from the reset state.  Lone re-runs (round 3) need one round per segment of the burst; the chain
    walk (round 4: the first listed segment of a run does them all, from exact state to exact state) one.  Both are exact."""
    cu8 = wm.synth_capture(seed=4242, n_samples=1 << 18, kinds=8, frames_per_s=40.0, amplitude=60.0, l_min=60, l_max=120)[0]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    rounds = {}
    for chains in (False, True):
        got, reruns = run_emulated(emu, ref["bit"], [ref["m"]], 1024, 64, chains=chains)
        rounds[chains] = ctypes.c_int.in_dll(emu, "wm_emu_last_rounds").value
        assert reruns > 0
        for ch in (0, 1):
            assert np.array_equal(got[ch], truncate_runs(oracle_rla_chips(ref, ch))), (chains, ch)
    assert rounds[False] >= 4 and rounds[True] <= 2, rounds


def test_division_through_an_approximate_reciprocal_is_exact(tmp_path):
    """The run-length kernel's divisions (chips per run, bit-length update, samples per bit) go through v_rcp_f32 -- within 1 ulp,
    not correctly rounded -- and one +-1 repair (wm_udiv): tests/sdiv_check.cpp runs the formula with the reciprocal pushed 1 ulp
    either way over the whole divisor domain (10^9 cases)."""
    exe = str(tmp_path / "sdiv_check")
    subprocess.run(["g++", "-O2", "-o", exe, os.path.join(HERE, "sdiv_check.cpp")], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith(" 0 wrong"), p.stdout


def test_chip_flood_continues_in_the_spill_arena(emu, oracle, wm):
    """The reference's chip loop never gives up (rtl_wmbus.c:765-779); a segment that outgrows its primary region (the
    product's half a chip per sample) continues in chunks of the spill arena, re-runs reuse the segment's chain, and the
    chip stream read back through the chain is the oracle's, chip for chip."""
    cu8 = interferer_capture(wm)
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    oc = oracle_rla_chips(ref, 0)
    assert np.bincount(oc[:, 0] // 8192).max() > 2 * 8192                     # the oracle really emits more than two chips per sample
    for seg_len, lookback, pushes in ((8192, 1024, [ref["m"]]), (2048, 64, [ref["m"] // 3 // 2048 * 2048, ref["m"] - ref["m"] // 3 // 2048 * 2048])):
        got, reruns, warn = run_emulated(emu, ref["bit"], pushes, seg_len, lookback, cap=seg_len // 2 + 8, spill_words=1 << 20)
        assert warn == 0
        for ch in (0, 1):
            assert np.array_equal(got[ch], truncate_runs(oracle_rla_chips(ref, ch))), (seg_len, ch)
    assert reruns > 0


def test_exhausted_spill_arena_drops_chips_but_keeps_the_framer_exact(emu, oracle, wm):
    """Arena too small for the flood: the segment's surplus chips are dropped and the warning bit is raised -- nothing
    fails, every carried state is still exact, so everything outside the starved segments (and the next push) is still
    the oracle's chip stream."""
    cu8 = interferer_capture(wm)
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    M = ref["m"]; half = M // 2 // 8192 * 8192
    got, _, warn = run_emulated(emu, ref["bit"], [half, M - half], 8192, 1024, cap=8192 // 2 + 8, spill_words=2 * 2048)
    assert warn == 8
    for ch in (0, 1):
        want = truncate_runs(oracle_rla_chips(ref, ch))
        g, k = got[ch], 0
        # got is a subsequence of want: whole tails of segments are missing, nothing else
        seg_of = lambda a: a[:, 0] // 8192
        lost = 0
        for sg in np.unique(seg_of(want)):
            w, h = want[seg_of(want) == sg], g[seg_of(g) == sg]
            assert len(h) <= len(w) and np.array_equal(h, w[:len(h)]), (ch, sg)
            lost += len(w) - len(h)
        assert (lost > 0) == (ch == 0)
