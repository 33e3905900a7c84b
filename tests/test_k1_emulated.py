"""The demodulation kernel's DEVICE SOURCE (rtl-wmbus_amd/csrc/wm_k1_demod.h) compiled for the host with
clang++ and run block by block on a coroutine block emulator (tests/emu/block_emu.h, k1_emu.cpp): every
thread of a block is a coroutine, __syncthreads and the wave ballot are scheduling points.  Soft symbols
bit for bit and RSSI bytes against the oracle, for every compiled decimation, with and without the
+-325 kHz shift, the -a discriminator, several pushes, and the RSSI-filter repair path.  No GPU needed."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from cases import flags_to_oracle_opts

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "rtl-wmbus_amd", "csrc")
SO = os.path.join(HERE, "emu", "libk1_emu.so")
SRC = os.path.join(HERE, "emu", "k1_emu.cpp")
CLANG = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"      # needs clang (ext_vector_type)
HIST, SLACK = 4096, 256
F_SHIFT, F_ACCURATE, F_T1C1, F_S1 = 1, 2, 8, 16                        # WM_F_* of wm_dev.h
FS = {1: 800, 2: 1600, 3: 2400, 4: 3200, 5: 4000, 6: 4800, 8: 6400}


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ for the host build of the device source")
    deps = [SRC, os.path.join(HERE, "emu", "block_emu.h")] + [os.path.join(CSRC, f) for f in ("wm_k1_demod.h", "wm_dev.h", "wm_exact.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run([CLANG, "-O1", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + CSRC, "-I" + os.path.join(HERE, "emu"),
                        "-Wno-unknown-pragmas", "-Wno-pass-failed", "-o", SO, SRC], check=True)
    L = ctypes.CDLL(SO)
    L.wm_emu_k1.restype = ctypes.c_long
    L.wm_emu_k1.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint,
                            ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return L


def run_emulated(emu, cu8, d, flags, push_bytes, polyphase=0):
    """One capture through K1 push by push.  Returns (dphi [2][M] float32, rssi [2][M] uint8, repaired tiles)."""
    total = cu8.size // 4096 * 4096
    max_push = max(push_bytes)
    stride = (HIST + max_push + SLACK + 255) // 256 * 256
    row = np.full(stride, 128, np.uint8)                   # wmbus_open fills the window with 128 (= zero input)
    carry = np.zeros(2, np.float32)
    out_d, out_r, n0, off, repaired = [[], []], [[], []], 0, 0, 0
    k = 0
    while off < total:
        nb = min(push_bytes[k % len(push_bytes)], total - off)
        k += 1
        row[HIST:HIST + nb] = cu8[off:off + nb]
        n_new = nb // 2
        M = (n0 + n_new) // d - n0 // d
        ntiles = (M + 975) // 976
        Mcap = max(256, (ntiles * 976 + 255) // 256 * 256)
        dphi = np.zeros((2, Mcap), np.float32)
        rssi = np.zeros((2, Mcap), np.uint8)
        err = ctypes.c_uint(0)
        r = emu.wm_emu_k1(row.ctypes.data, stride, 1, d, flags, n0, n_new, Mcap, dphi.ctypes.data, rssi.ctypes.data, carry.ctypes.data, ctypes.byref(err), polyphase)
        assert r >= 0 and err.value == 0
        repaired += r
        for ch in range(2):
            out_d[ch].append(dphi[ch, :M].copy()); out_r[ch].append(rssi[ch, :M].copy())
        row[:HIST] = row[nb:nb + HIST].copy()              # k_roll_history
        n0 += n_new; off += nb
    return [np.concatenate(x) for x in out_d], [np.concatenate(x) for x in out_r], repaired


def check(emu, oracle, cu8, flags_cli, d, pushes, polyphase=0, atan_mode=0):
    oo = flags_to_oracle_opts(oracle, flags_cli)
    oo.prefilter = polyphase
    oo.atan_mode = atan_mode
    ref = oracle.run(cu8, oo, taps=True)
    flags = F_T1C1 | F_S1 | (F_SHIFT if "-s" in flags_cli else 0) | (0 if "-a" in flags_cli else F_ACCURATE) | (128 * atan_mode)   # -a = the fast discriminator; 128 / 256 = WM_F_APPROX1 / 2
    dphi, rssi, repaired = run_emulated(emu, cu8, d, flags, pushes, polyphase)
    for ch in (0, 1):
        assert len(dphi[ch]) == ref["m"]
        assert np.array_equal(dphi[ch].view(np.uint32), ref["dphi_fir"][ch].view(np.uint32)), ("dphi", ch, flags_cli)
        assert np.array_equal(rssi[ch], ref["rssi"][ch].astype(np.uint32).astype(np.uint8)), ("rssi", ch, flags_cli)
    return repaired


@pytest.mark.parametrize("flags_cli", [["-v"], ["-v", "-a"], ["-v", "-s"]])
def test_device_source_on_host_matches_oracle_bundled_capture(emu, oracle, samples, flags_cli):
    cu8 = samples["samples2"][: 1 << 19]                   # a quarter of the capture: the emulation runs ~0.3 M samples/s
    check(emu, oracle, cu8, flags_cli, 2, [cu8.size])


@pytest.mark.parametrize("d,extra", [(3, []), (4, ["-s"]), (5, ["-s"]), (6, []), (2, [])])
def test_device_source_on_host_matches_oracle_decimations_and_pushes(emu, oracle, wm, d, extra):
    kw = dict(t1c1_center_khz=325.0, s1_center_khz=-325.0) if "-s" in extra else {}
    cu8 = wm.synth_capture(seed=400 + d, n_samples=3 << 16, fs_khz=FS[d], kinds=15, frames_per_s=150.0, amplitude=40.0, **kw)[0]
    flags_cli = ["-v"] + (["-d", str(d)] if d != 2 else []) + extra
    check(emu, oracle, cu8, flags_cli, d, [4096 * 7, 4096 * 20, 4096, 4096 * 64])


def test_rssi_filter_repair_path_on_host(emu, oracle, wm):
    """Signal, then exact silence: the EMA decays through ~90 samples of subnormals while a warm-up from
    zero is already at zero -- hand-offs fail certification and are repaired sequentially."""
    cu8 = wm.synth_capture(seed=77, n_samples=1 << 17, kinds=15, frames_per_s=400.0, amplitude=60.0)[0]
    cu8[cu8.size // 2:] = 128
    repaired = check(emu, oracle, cu8, ["-v"], 2, [cu8.size])
    assert repaired > 0


def test_polyphase_prefilter_kernel_on_host(emu, oracle, samples):
    cu8 = samples["samples2"][: 1 << 19]
    check(emu, oracle, cu8, ["-v"], 2, [4096 * 33, 4096 * 5], polyphase=1)


def test_randomised_captures_on_host(emu, oracle, wm):
    rng = np.random.default_rng(3 + int(os.environ.get("WMBUS_EMU_SEED", "0")))
    for k in range(int(os.environ.get("WMBUS_EMU_N", "4"))):                 # more for a bug hunt
        d = int(rng.choice([2, 2, 3, 4, 5, 6]))
        shift = bool(rng.random() < 0.3)
        kw = dict(t1c1_center_khz=325.0, s1_center_khz=-325.0) if shift else {}
        cu8 = wm.synth_capture(seed=int(rng.integers(1, 1 << 30)), n_samples=int(rng.choice([1 << 16, 3 << 15])), fs_khz=FS[d], kinds=15,
                               frames_per_s=200.0, amplitude=float(rng.choice([8.0, 60.0, 120.0])), noise_sigma=float(rng.choice([0.0, 3.0, 20.0])), **kw)[0]
        if k % 2:
            a = int(rng.integers(0, cu8.size // 2)) & ~1
            cu8[a:a + int(rng.integers(4096, cu8.size // 2))] = int(rng.choice([0, 127, 128, 255]))
        flags_cli = ["-v"] + (["-d", str(d)] if d != 2 else []) + (["-s"] if shift else []) + (["-a"] if rng.random() < 0.2 else [])
        check(emu, oracle, cu8, flags_cli, d, [int(x) * 4096 for x in rng.integers(1, 30, 4)])


@pytest.mark.parametrize("atan_mode", [1, 2])
@pytest.mark.parametrize("d,extra,polyphase", [(2, [], 0), (3, ["-s"], 0), (2, [], 1)])
def test_atan2_approximation_options_on_host(emu, oracle, wm, atan_mode, d, extra, polyphase):
    """wmbus_cfg.atan_mode: atan2.h's two approximations in place of cargf (options of the reference's source,
    never compiled into its binary): the kernel against the oracle, whose functions are pinned to the
    reference's own (tests/test_oracle_units.py)."""
    kw = dict(t1c1_center_khz=325.0, s1_center_khz=-325.0) if "-s" in extra else {}
    cu8 = wm.synth_capture(seed=900 + d + atan_mode, n_samples=1 << 16, fs_khz=FS[d], kinds=15, frames_per_s=200.0, amplitude=50.0, **kw)[0]
    cu8[cu8.size // 2: cu8.size // 2 + 20000] = 128                       # exact silence: x = y = 0
    check(emu, oracle, cu8, ["-v"] + (["-d", str(d)] if d != 2 else []) + extra, d, [4096 * 9, 4096 * 2], polyphase, atan_mode)


def run_on_demand(emu, cu8, d, flags, push_bytes, rng, density, big=False, tpb=1):
    """One capture push by push the RSSI-on-demand way: the first pass without RSSI, then the RSSI of randomly flagged tiles.
    Returns dphi [2][M], rssi [2][M], read [2][M] bool (samples whose RSSI was asked for), the filter state per push, failures."""
    emu.wm_emu_k1_od.restype = ctypes.c_long
    emu.wm_emu_k1_od.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, ctypes.c_uint,
                                 ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    ctypes.c_int.in_dll(emu, "wm_emu_k1_big").value = int(big)    # the first pass on 2000-sample tiles of 512 threads (wmbus_ctx.k1_big)
    ctypes.c_int.in_dll(emu, "wm_emu_k1_tpb").value = int(tpb)    # ... several consecutive tiles per block, the next tile's input prefetched (K1Args.tpb)
    total = cu8.size // 4096 * 4096
    stride = (HIST + max(push_bytes) + 2 * (2048 + 16) * d + SLACK + 255) // 256 * 256     # a partial last tile reads (never uses) a tile past the staged bytes
    row = np.full(stride, 128, np.uint8)
    out_d, out_r, out_m, states, n0, off, k, fails = [[], []], [[], []], [[], []], [], 0, 0, 0, 0
    while off < total:
        nb = min(push_bytes[k % len(push_bytes)], total - off)
        k += 1
        row[HIST:HIST + nb] = cu8[off:off + nb]
        n_new = nb // 2
        M = (n0 + n_new) // d - n0 // d
        ntiles = (M + 975) // 976
        Mcap = max(256, (ntiles * 976 + 255) // 256 * 256)
        dphi = np.zeros((2, Mcap), np.float32); rssi = np.full((2, Mcap), 0xEE, np.uint8); state = np.zeros(2, np.float32)
        tf = (rng.random(ntiles) < density).astype(np.uint32) * rng.integers(1, 4, ntiles).astype(np.uint32)
        r = emu.wm_emu_k1_od(row.ctypes.data, stride, 1, d, flags, n0, n_new, Mcap, dphi.ctypes.data, rssi.ctypes.data, state.ctypes.data, tf.ctypes.data)
        assert r in (0, 1), r
        fails += r
        for ch in range(2):
            read = np.repeat((tf >> ch) & 1, 976)[:M].astype(bool)
            out_d[ch].append(dphi[ch, :M].copy()); out_r[ch].append(rssi[ch, :M].copy()); out_m[ch].append(read)
        states.append((n0 // d + M - 1, state.copy(), r))
        row[:HIST] = row[nb:nb + HIST].copy()
        n0 += n_new; off += nb
    return [np.concatenate(x) for x in out_d], [np.concatenate(x) for x in out_r], [np.concatenate(x) for x in out_m], states, fails


@pytest.mark.parametrize("d,flags_cli,big,tpb", [(2, ["-v"], False, 1), (2, ["-v"], True, 1), (2, ["-v"], False, 2), (2, ["-v"], True, 3), (2, ["-v", "-s"], False, 2),
                                                 (3, ["-v", "-d", "3"], False, 4), (5, ["-v", "-d", "5", "-s"], False, 1), (4, ["-v", "-d", "4"], False, 64)])
def test_rssi_on_demand_matches_the_oracle_where_it_is_read(emu, oracle, wm, d, flags_cli, big, tpb):
    """RS = 1 leaves the soft symbols as they were; RS = 2 fills in the RSSI of the flagged tiles, every lane proving the
    state it starts from by a bracket of two trajectories (wm_k1_demod.h), and hands on the filter's state.  big: the first
    pass runs on 2000-sample tiles of 512 threads (an option at decimation 2 without -s), the RSSI launch on its 976; tpb: a block of the
    first pass takes that many consecutive tiles, the input of the next one loaded while the current one is computed."""
    rng = np.random.default_rng(77 + d)
    cu8 = wm.synth_capture(seed=4242 + d, n_samples=1 << 17, kinds=15, frames_per_s=200.0, fs_khz=FS[d])[0]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, flags_cli), taps=True)
    flags = F_ACCURATE | F_T1C1 | F_S1 | (F_SHIFT if "-s" in flags_cli else 0)
    for pushes in ([1 << 18], [4096 * 5, 4096 * 16, 4096]):
        dphi, rssi, read, states, fails = run_on_demand(emu, cu8, d, flags, pushes, rng, 0.3, big=big, tpb=tpb)
        assert fails == 0                                   # noise and signal: every bracket closes inside the warm-up
        for ch in (0, 1):
            m = len(dphi[ch])
            assert np.array_equal(dphi[ch].view(np.uint32), ref["dphi_fir"][ch][:m].view(np.uint32)), ("soft symbols", ch)
            want = ref["rssi"][ch][:m].astype(np.uint32).astype(np.uint8)
            assert read[ch].any() and np.array_equal(rssi[ch][read[ch]], want[read[ch]]), ("rssi", ch)
            assert np.all(rssi[ch][~read[ch]] == 0xEE)      # nothing else was computed
        for last, state, _ in states:
            for ch in (0, 1):
                assert state[ch].view(np.uint32) == np.float32(ref["rssi"][ch][last]).view(np.uint32), ("carried state", ch, last)


def test_rssi_on_demand_reports_what_it_cannot_prove(emu, oracle, wm):
    """Exact silence after a signal: constant input can keep the bracket open (the filter step has neighbouring fixed
    points), and the true state decays through a hundred samples while a warm-up from zero is there at once.  The launch
    must say so (the product then runs the full pass) -- and whatever it does NOT report must be right."""
    rng = np.random.default_rng(5)
    cu8 = wm.synth_capture(seed=99, n_samples=1 << 17, kinds=15, frames_per_s=200.0, amplitude=60.0)[0]
    cu8[2 * 30000:2 * 60000] = 128                           # exact zero input
    cu8[2 * 80000:2 * 100000:2] = 127; cu8[2 * 80000 + 1:2 * 100000:2] = 128      # constant, not zero
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True)
    flags = F_ACCURATE | F_T1C1 | F_S1
    seen_fail = 0
    for pushes in ([1 << 18], [4096 * 7], [4096 * 3, 4096 * 11]):
        dphi, rssi, read, states, fails = run_on_demand(emu, cu8, 2, flags, pushes, rng, 1.0)
        seen_fail += fails
        off = 0
        for last, state, failed in states:
            if not failed:
                for ch in (0, 1):
                    want = ref["rssi"][ch][off:last + 1].astype(np.uint32).astype(np.uint8)
                    sel = read[ch][off:last + 1]
                    assert np.array_equal(rssi[ch][off:last + 1][sel], want[sel]), ("rssi", ch, off, last)
                    assert state[ch].view(np.uint32) == np.float32(ref["rssi"][ch][last]).view(np.uint32)
            off = last + 1
    assert seen_fail > 0
