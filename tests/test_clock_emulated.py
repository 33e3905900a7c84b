"""The clock-recovery / time2 framer lanes' DEVICE SOURCE (rtl-wmbus_amd/csrc/wm_k2_clock.h, and its systolic form
wm_k2_clock_sys.h: a lane group on the four waves of a block) compiled for the host and run lane by lane / block by
block (tests/emu/clock_emu.cpp) against the oracle: slicer bits and time2 chips, with speculative cold starts,
hand-off verification, re-run rounds, checkpoints / early exit and the carry across pushes.  No GPU needed; the GPU
suite checks the compiled kernel (and its cooperative loads)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from cases import flags_to_oracle_opts

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "rtl-wmbus_amd", "csrc")
SO = os.path.join(HERE, "emu", "libclock_emu.so")
SRC = os.path.join(HERE, "emu", "clock_emu.cpp")
F_DC, F_T1C1, F_S1, F_T2A = 4, 8, 16, 64                   # WM_F_* of wm_dev.h


@pytest.fixture(scope="module")
def emu():
    deps = [SRC, os.path.join(HERE, "emu", "block_emu.h")] + [os.path.join(CSRC, f) for f in ("wm_k2_clock.h", "wm_k2_clock_sys.h", "wm_k2_sys_blocks.h", "wm_k2_common.h", "wm_dev.h", "wm_exact.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + CSRC, "-I" + os.path.join(HERE, "emu"),
                        "-Wno-unknown-pragmas", "-o", SO, SRC], check=True)
    L = ctypes.CDLL(SO)
    L.wm_emu_clock.restype = ctypes.c_long
    L.wm_emu_clock.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 8 + [ctypes.c_void_p] * 6
    L.wm_emu_clock_state_bytes.restype = ctypes.c_uint
    return L


def run_emulated(emu, soft_rows, pushes, seg_len, warm, dc=False, descending=True, s1_span=None, chains=True, sys=False, states=None):
    """soft_rows: [2][M] float32 FIR outputs of one capture; pushes: decimated samples per push.  sys: the systolic form (four
    coroutines per lane on the block emulator).  states: a list that receives every push's start / end state records."""
    ctypes.c_int.in_dll(emu, "wm_emu_sys").value = int(sys)
    if sys:
        s1_span = 1                                          # the systolic form has no two-segment S1 lanes (the product never used them)
    ctypes.c_int.in_dll(emu, "wm_emu_descending").value = int(descending)   # see clock_emu.cpp: launch semantics
    ctypes.c_int.in_dll(emu, "wm_emu_s1_span").value = int(s1_span or 0)     # WmPush.s1_span: S1 lanes cover two segments
    ctypes.c_int.in_dll(emu, "wm_emu_chains").value = int(chains)            # a listed lane walks its chain of listed segments (K2Args.bad)
    sb = emu.wm_emu_clock_state_bytes()
    carry = np.zeros(2 * sb, np.uint8)                     # a fresh context starts from the all-zero state
    chips_out, bits_out, m0, reruns, max_rounds = [[], []], [[], []], 0, 0, 0
    cap = seg_len // 4 + 8
    flags = F_T1C1 | F_S1 | F_T2A | (F_DC if dc else 0)
    for M in pushes:
        Mcap = (M + 255) // 256 * 256
        x = np.zeros((2, Mcap), np.float32)
        for ch in range(2):
            x[ch, :M] = soft_rows[ch][m0:m0 + M]
        nseg = (M + seg_len - 1) // seg_len
        bits = np.zeros((2, Mcap // 32), np.uint32)
        chips = np.zeros((2, nseg, cap), np.uint32)
        counts = np.zeros((2, nseg), np.uint32)
        err, rounds = ctypes.c_uint(0), ctypes.c_uint(0)
        st = np.zeros(2 * 2 * nseg * sb, np.uint8)
        ctypes.c_void_p.in_dll(emu, "wm_emu_states_out").value = st.ctypes.data if states is not None else None
        r = emu.wm_emu_clock(x.ctypes.data, 1, M, Mcap, flags, seg_len, warm[0], warm[1], cap, carry.ctypes.data, bits.ctypes.data,
                             chips.ctypes.data, counts.ctypes.data, ctypes.byref(err), ctypes.byref(rounds))
        ctypes.c_void_p.in_dll(emu, "wm_emu_states_out").value = None
        assert r >= 0 and err.value == 0
        if states is not None:
            states.append((st.copy(), counts.copy()))
        reruns += r
        max_rounds = max(max_rounds, rounds.value)
        for ch in range(2):
            bits_out[ch].append(np.unpackbits(bits[ch].view(np.uint8), bitorder="little")[:M])
            for s in range(nseg):
                w = chips[ch].reshape(-1)[s * cap: s * cap + counts[ch, s]]       # an S1 lane spanning two segments fills both regions from the first
                chips_out[ch].append(np.stack([m0 + s * seg_len + (w >> 3), w & 7], axis=1))
        m0 += M
    return ([np.concatenate(o) if o else np.zeros((0, 2), np.uint32) for o in chips_out], [np.concatenate(b) for b in bits_out],
            reruns, max_rounds)


def oracle_t2a_chips(ref, ch):
    oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == 1)]
    return np.stack([oc["sample"].astype(np.uint32), oc["value"].astype(np.uint32)], axis=1)


@pytest.mark.parametrize("s1_span", [1, 2, "sys"])
@pytest.mark.parametrize("seg_len,warm,flags", [(32768, (12288, 24576), ["-v"]), (4096, (512, 512), ["-v"]), (8192, (1024, 2048), ["-v", "-o"]),
                                                 (2048, (128, 256), ["-v"])])
def test_device_source_on_host_matches_oracle_bundled_capture(emu, oracle, samples, seg_len, warm, flags, s1_span):
    cu8 = samples["samples2"]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, flags), taps=True, chips=True)
    chips, bits, reruns, rounds = run_emulated(emu, ref["dphi_fir"], [ref["m"]], seg_len, warm, dc="-o" in flags, s1_span=1 if s1_span == "sys" else s1_span, sys=s1_span == "sys")
    for ch in (0, 1):
        assert np.array_equal(bits[ch], ref["bit"][ch]), ("bits", ch)
        assert np.array_equal(chips[ch], oracle_t2a_chips(ref, ch)), ("chips", ch)
    if seg_len <= 4096:
        assert reruns > 0


def test_device_source_on_host_matches_oracle_randomised(emu, oracle, wm):
    rng = np.random.default_rng(9 + int(os.environ.get("WMBUS_EMU_SEED", "0")))
    multi, walked, sys_reruns = {False: 0, True: 0}, 0, 0
    for k in range(int(os.environ.get("WMBUS_EMU_N", "8"))):               # more for a bug hunt
        cu8 = wm.synth_capture(seed=int(rng.integers(1, 1 << 30)), n_samples=1 << 18, kinds=int(rng.choice([15, 15, 8, 7])), frames_per_s=120.0,
                               amplitude=float(rng.choice([8.0, 25.0, 60.0])), noise_sigma=float(rng.choice([0.5, 3.0, 3.0, 10.0])))[0]
        if k % 3 == 1:
            a = int(rng.integers(0, cu8.size // 2)) & ~1
            cu8[a:a + int(rng.integers(4096, cu8.size // 3))] = int(rng.choice([127, 128]))
        dc = bool(k % 4 == 3)
        ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"] + (["-o"] if dc else [])), taps=True, chips=True)
        M = ref["m"]
        q = 1024 if k % 3 else 1
        cuts = sorted(set(int(x) // q * q for x in rng.integers(1024, M, 3)))
        pushes = [b - a for a, b in zip([0] + cuts, cuts + [M]) if b > a]
        seg_len = int(rng.choice([2048, 4096, 8192, 32768]))
        warm = (int(rng.choice([64, 128, 512, 4096, 12288])), int(rng.choice([64, 128, 512, 8192, 24576])))
        if os.environ.get("WMBUS_EMU_STRESS"):               # bug hunts: many checkpoints per segment, hopeless warm-ups
            seg_len = int(rng.choice([4096, 8192, 16384]))
            warm = (int(rng.choice([32, 64, 256])), int(rng.choice([32, 64, 256])))
        for chains in (False, True):                          # round 3's rounds of lone segments; round 4's chain walk
            st_one, st_sys = [], []
            chips, bits, reruns, rounds = run_emulated(emu, ref["dphi_fir"], pushes, seg_len, warm, dc=dc, descending=bool(k % 5),
                                                       s1_span=1 + (k // 2) % 2, chains=chains, states=st_one)
            multi[chains] += rounds > 1
            walked += chains and reruns > 0
            for ch in (0, 1):
                assert np.array_equal(bits[ch], ref["bit"][ch]), (k, "bits", ch, chains)
                assert np.array_equal(chips[ch], oracle_t2a_chips(ref, ch)), (k, "chips", ch, seg_len, warm, chains)
            # the systolic form (round 6): the same chips and slicer bits, and once every hand-off is certified the SAME records in memory
            chips, bits, reruns_s, rounds_s = run_emulated(emu, ref["dphi_fir"], pushes, seg_len, warm, dc=dc, descending=bool(k % 5), chains=chains, sys=True, states=st_sys)
            sys_reruns += reruns_s
            for ch in (0, 1):
                assert np.array_equal(bits[ch], ref["bit"][ch]), (k, "bits", ch, chains, "sys")
                assert np.array_equal(chips[ch], oracle_t2a_chips(ref, ch)), (k, "chips", ch, seg_len, warm, chains, "sys")
            if (k // 2) % 2 == 0:                             # (two-segment S1 lanes lay their records out differently)
                for (a, ca), (b, cb) in zip(st_one, st_sys):
                    assert np.array_equal(a, b) and np.array_equal(ca, cb), (k, "state records", chains)
    assert multi[False] > 0                                    # cascading re-run rounds (where the checkpoint bug lived) occurred
    assert walked > 0 and multi[True] <= multi[False]          # the chain walk ran, and never needs more rounds than lone segments do
    assert sys_reruns > 0


@pytest.mark.parametrize("sys", [0, 1])
def test_cooperative_first_pass_of_a_whole_wave_on_the_block_emulator(emu, oracle, wm, sys):
    """64 captures = one wave per (chain, segment): the first pass fetches the rows cooperatively (8 lanes per
    row, one 128-byte line each) and transposes the block through LDS between wave barriers.  The 64 lanes run
    as coroutines on the block emulator; re-runs take the lane-private path.  sys = 1: the systolic form -- role 0's wave
    loads cooperatively, 256 coroutines per block, lanes of a re-run chunk at different points of different segments;
    here with a short last segment that ends in a ragged tail."""
    ctypes.c_int.in_dll(emu, "wm_emu_sys").value = int(sys)
    ctypes.c_int.in_dll(emu, "wm_emu_s1_span").value = 1
    S, seg_len, warm = 64, 8192, (1024, 2048)
    if sys:
        seg_len, warm = 4096, (1024 + 64, 2048 + 32)          # warm-ups that end off a checkpoint grid, several segments, a short last one
    refs = []
    for s in range(S):
        cu8 = wm.synth_capture(seed=7000 + s, n_samples=1 << 16, kinds=15, frames_per_s=300.0, amplitude=60.0)[0]
        refs.append(oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True))
    M = refs[0]["m"] - (1024 + 13 if sys else 0)            # a short last segment with a ragged tail
    Mcap = (M + 255) // 256 * 256
    x = np.zeros((2, S, Mcap), np.float32)
    for s in range(S):
        for ch in range(2):
            x[ch, s, :M] = refs[s]["dphi_fir"][ch][:M]
    nseg, cap = (M + seg_len - 1) // seg_len, seg_len // 4 + 8
    bits = np.zeros((2, S, Mcap // 32), np.uint32); chips = np.zeros((2, S, nseg, cap), np.uint32); counts = np.zeros((2, S, nseg), np.uint32)
    carry = np.zeros(2 * S * emu.wm_emu_clock_state_bytes(), np.uint8); err = ctypes.c_uint(0); rounds = ctypes.c_uint(0)
    r = emu.wm_emu_clock(x.ctypes.data, S, M, Mcap, F_T1C1 | F_S1 | F_T2A, seg_len, warm[0], warm[1], cap, carry.ctypes.data, bits.ctypes.data,
                         chips.ctypes.data, counts.ctypes.data, ctypes.byref(err), ctypes.byref(rounds))
    assert r > 0 and err.value == 0                        # short warm-ups: the re-run path ran too
    for s in range(S):
        for ch in range(2):
            assert np.array_equal(np.unpackbits(bits[ch, s].view(np.uint8), bitorder="little")[:M], refs[s]["bit"][ch][:M]), ("bits", s, ch)
            got = np.concatenate([np.stack([g * seg_len + (chips[ch, s, g, :counts[ch, s, g]] >> 3), chips[ch, s, g, :counts[ch, s, g]] & 7], axis=1)
                                  for g in range(nseg)])
            want = oracle_t2a_chips(refs[s], ch)
            assert np.array_equal(got, want[want[:, 0] < M]), ("chips", s, ch)
