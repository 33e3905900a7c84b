"""k3_scan + k3_bursts (device source rtl-wmbus_amd/csrc/wm_k3_bursts.h) on the coroutine block emulator,
fed by the host-emulated framer kernels: the bursts the GPU would ship to the host packet decoders must be
exactly the oracle's chips after every access-code chip -- position, value and RSSI of each chip, burst
length from the L-field -- plus the continuation slots of decoders that were left busy.  No GPU needed."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from cases import flags_to_oracle_opts
import test_clock_emulated as CE
import test_rla_emulated as RE
import test_burst_need as BN

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "rtl-wmbus_amd", "csrc")
SO = os.path.join(HERE, "emu", "libk3_emu.so")
SRC = os.path.join(HERE, "emu", "k3_emu.cpp")
CLANG = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"
HDR = np.dtype([("stream", "<u4"), ("chain", "u1"), ("algo", "u1"), ("flags", "<u2"), ("chip0", "<u4"), ("n_chips", "<u4"),
                ("pos0", "<u8"), ("word_off", "<u4"), ("avail", "<u4")])
PKT = np.dtype([("stream", "<u4"), ("chain", "u1"), ("algo", "u1"), ("status", "u1"), ("flags", "u1"), ("chip0", "<u4"), ("consumed", "<u4"),
                ("sample", "<u8"), ("off", "<u4"), ("L", "<u2"), ("pkt_rssi", "u1"), ("rssi_now", "u1")])
F_T1C1, F_S1, F_RLA, F_T2A = 8, 16, 32, 64

emu_clock = CE.emu
emu_rla = RE.emu
emu_need = BN.emu


@pytest.fixture(scope="module")
def emu_k3():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++")
    deps = [SRC, os.path.join(HERE, "emu", "block_emu.h")] + [os.path.join(CSRC, f) for f in ("wm_k3_bursts.h", "wm_k2_common.h", "wm_dev.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run([CLANG, "-O1", "-std=c++17", "-shared", "-fPIC", "-I" + CSRC, "-I" + os.path.join(HERE, "emu"), "-Wno-unknown-pragmas",
                        "-o", SO, SRC], check=True)
    L = ctypes.CDLL(SO)
    L.wm_emu_k3.restype = ctypes.c_long
    L.wm_emu_k3.argtypes = [ctypes.c_void_p] * 10 + [ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint]
    L.wm_emu_k3_set_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    L.wm_emu_k3_set_spill.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert L.wm_emu_hdr_bytes() == HDR.itemsize and L.wm_emu_pkt_bytes() == PKT.itemsize
    return L


def framers_on_host(emu_clock, emu_rla, ref, seg1=32768, seg0=8192):
    """One push through the emulated clock and run-length kernels; returns their raw region arrays."""
    M = ref["m"]; Mcap = (M + 255) // 256 * 256
    x = np.zeros((2, Mcap), np.float32)
    for ch in range(2):
        x[ch, :M] = ref["dphi_fir"][ch]
    out = {}
    # clock / time2
    nseg1, cap1 = (M + seg1 - 1) // seg1, seg1 // 4 + 8
    bits = np.zeros((2, Mcap // 32), np.uint32); chips1 = np.zeros((2, nseg1, cap1), np.uint32); counts1 = np.zeros((2, nseg1), np.uint32)
    seen1 = np.zeros((2, nseg1), np.uint32); carry1 = np.zeros(2 * emu_clock.wm_emu_clock_state_bytes(), np.uint8); err = ctypes.c_uint(0)
    ctypes.c_void_p.in_dll(emu_clock, "wm_emu_seen_out").value = seen1.ctypes.data
    r = emu_clock.wm_emu_clock(x.ctypes.data, 1, M, Mcap, F_T1C1 | F_S1 | F_T2A, seg1, 4096, 8192, cap1, carry1.ctypes.data, bits.ctypes.data,
                               chips1.ctypes.data, counts1.ctypes.data, ctypes.byref(err), None)
    ctypes.c_void_p.in_dll(emu_clock, "wm_emu_seen_out").value = None
    assert r >= 0 and err.value == 0
    # run-length
    nseg0, cap0 = (M + seg0 - 1) // seg0, 4 * seg0 + 8 + 8192
    chips0 = np.zeros((2, nseg0, cap0), np.uint32); counts0 = np.zeros((2, nseg0), np.uint32); seen0 = np.zeros((2, nseg0), np.uint32)
    sb = emu_rla.wm_emu_rla_state_bytes(); carry0 = np.zeros(2 * sb, np.uint8)
    for rr in range(2):
        emu_rla.wm_emu_rla_reset_state(carry0[rr * sb:].ctypes.data)
    ctypes.c_void_p.in_dll(emu_rla, "wm_emu_seen_out").value = seen0.ctypes.data
    emu_rla.wm_emu_rla_set_spill(None, 0, None, None, None)             # roomy primary regions here
    r = emu_rla.wm_emu_rla(bits.ctypes.data, 1, M, Mcap, F_T1C1 | F_S1, seg0, 1024, cap0, carry0.ctypes.data, chips0.ctypes.data, counts0.ctypes.data,
                           ctypes.byref(err))
    ctypes.c_void_p.in_dll(emu_rla, "wm_emu_seen_out").value = None
    assert r >= 0 and err.value == 0
    geo = np.array([M, Mcap, F_T1C1 | F_S1 | F_RLA | F_T2A, 0, seg0, seg1, nseg0, nseg1, cap0, cap1], np.uint64)
    return dict(geo=geo, chips=(chips0, chips1), counts=(counts0, counts1), seen=(seen0, seen1), Mcap=Mcap)


def bursts_on_host(emu_k3, fr, rssi_rows, pending=None, max_blocks=256, decode=False, spans=True):
    """decode=False: every burst as chips for the host decoders (hdr, words).  decode=True: as the product runs it --
    bursts that end inside the push are decoded by the kernel: (hdr, words, pkts, bytes).
    spans (on in every test): RSSI on demand -- k3_spans flags the demodulation tiles whose RSSI the bursts read, and
    k3_bursts then sees zeros in every other tile (k3_emu.cpp): what comes out must not change.  The flags of the last call
    are kept in bursts_on_host.last_spans."""
    emu_k3.wm_emu_k3_set_spans.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    flags = np.zeros((int(fr["geo"][0]) + 975) // 976, np.uint32) if spans else None
    emu_k3.wm_emu_k3_set_spans(flags.ctypes.data if spans else None, flags.size if spans else 0)
    bursts_on_host.last_spans = flags
    pend = np.zeros(4, np.uint32) if pending is None else np.asarray(pending, np.uint32)
    hdr = np.zeros(1 << 16, HDR); words = np.zeros(1 << 24, np.uint32); nw = ctypes.c_uint(0)
    emu_k3.wm_emu_k3_set_spill(*fr.get("spill", (None, 0, None, None, None)))
    pkts = np.zeros(1 << 16, PKT); pbytes = np.zeros(1 << 22, np.uint8); npk = ctypes.c_uint(0)
    if decode:
        emu_k3.wm_emu_k3_set_decode(pkts.ctypes.data, pkts.size, ctypes.addressof(npk), pbytes.ctypes.data, pbytes.size)
    else:
        emu_k3.wm_emu_k3_set_decode(None, 0, None, None, 0)
    n = emu_k3.wm_emu_k3(fr["geo"].ctypes.data, fr["chips"][0].ctypes.data, fr["chips"][1].ctypes.data, fr["counts"][0].ctypes.data,
                         fr["counts"][1].ctypes.data, fr["seen"][0].ctypes.data, fr["seen"][1].ctypes.data, rssi_rows.ctypes.data,
                         pend.ctypes.data, hdr.ctypes.data, hdr.size, words.ctypes.data, words.size, ctypes.byref(nw), max_blocks)
    assert n >= 0, n
    emu_k3.wm_emu_k3_set_decode(None, 0, None, None, 0)
    emu_k3.wm_emu_k3_set_spans(None, 0)
    if decode:
        return hdr[:n], words[:nw.value], pkts[:npk.value], pbytes
    return hdr[:n], words[:nw.value]


def expected_bursts(emu_need, ref, ch, al):
    oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al)]
    pos, val, rssi = oc["sample"].astype(np.int64), oc["value"].astype(np.uint32), oc["rssi"].astype(np.uint32)
    out = {}
    for i in np.nonzero(val & 2)[0]:
        avail = len(val) - i
        nb = min(24, avail - 1)
        hb = 0
        for b in val[i + 1:i + 1 + nb] & 1:
            hb = (hb << 1) | int(b)
        hb <<= 24 - nb
        n = min(emu_need.wm_emu_burst_need(ch, hb, nb) + 1, avail)
        w = ((pos[i:i + n] - pos[i]).astype(np.uint32) << 11) | (rssi[i:i + n] << 3) | (val[i:i + n] & 7)
        out[int(i)] = (n, int(pos[i]), avail, w)
    return out


CASES = [(1, 60.0, 15, 32768, 8192), (2, 25.0, 15, 32768, 8192)] + [
    (100 + k + 1000 * int(os.environ.get("WMBUS_EMU_SEED", "0")), [8.0, 25.0, 60.0][k % 3], [15, 8, 7][k % 3], [4096, 8192, 32768][k % 3], [1024, 8192][k % 2])
    for k in range(int(os.environ.get("WMBUS_EMU_N", "2")))]                      # more for a bug hunt


@pytest.mark.parametrize("seed,amp,kinds,seg1,seg0", CASES)
def test_bursts_are_the_oracles_chips_after_every_access_code(emu_k3, emu_clock, emu_rla, emu_need, oracle, wm, seed, amp, kinds, seg1, seg0):
    cu8 = wm.synth_capture(seed=5000 + seed, n_samples=1 << 19, kinds=kinds, frames_per_s=150.0, amplitude=amp)[0]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    fr = framers_on_host(emu_clock, emu_rla, ref, seg1, seg0)
    rssi = np.zeros((2, fr["Mcap"]), np.uint8)
    for ch in range(2):
        rssi[ch, :ref["m"]] = ref["rssi"][ch].astype(np.uint32).astype(np.uint8)
    hdr, words = bursts_on_host(emu_k3, fr, rssi)
    assert not (hdr["flags"] & 1).any()                    # nothing pending: no continuation bursts
    sp = bursts_on_host.last_spans                          # RSSI on demand: the tiles k3_spans listed; every other tile read as zero just now
    print("tiles listed per chain: %d / %d of %d" % (int((sp & 1).astype(bool).sum()), int((sp & 2).astype(bool).sum()), sp.size))
    assert sp[-1] == 3 and (sp & 3).astype(bool).any() and not ((sp & 1).astype(bool).all() and (sp & 2).astype(bool).all())
    n_checked = 0
    for ch in (0, 1):
        for al in (0, 1):
            want = expected_bursts(emu_need, ref, ch, al)
            got = hdr[(hdr["chain"] == ch) & (hdr["algo"] == al)]
            assert sorted(got["chip0"].tolist()) == sorted(want), (ch, al)
            for h in got:
                n, pos0, avail, w = want[int(h["chip0"])]
                assert (h["n_chips"], h["pos0"], h["avail"]) == (n, pos0, avail), (ch, al, h)
                assert np.array_equal(words[h["word_off"]:h["word_off"] + n], w), (ch, al, h)
                n_checked += 1
    assert n_checked > 5


def test_continuation_slots_deliver_what_a_busy_decoder_is_owed(emu_k3, emu_clock, emu_rla, oracle, wm):
    cu8 = wm.synth_capture(seed=5100, n_samples=1 << 18, kinds=15, frames_per_s=150.0, amplitude=60.0)[0]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    fr = framers_on_host(emu_clock, emu_rla, ref)
    rssi = np.zeros((2, fr["Mcap"]), np.uint8)
    pending = [0, 700, 33, 10 ** 7]                        # [algo][chain]: rla/T1C1 none, rla/S1 700, t2a/T1C1 33, t2a/S1 more than there is
    hdr, words = bursts_on_host(emu_k3, fr, rssi, pending)
    cont = hdr[(hdr["flags"] & 1) == 1]
    assert len(cont) == 3
    for h in cont:
        al, ch = int(h["algo"]), int(h["chain"])
        oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al)]
        assert h["chip0"] == 0 and h["n_chips"] == min(pending[al * 2 + ch], len(oc)) and h["avail"] == len(oc)
        w = words[h["word_off"]:h["word_off"] + h["n_chips"]]
        assert np.array_equal(w & 7, oc["value"][:h["n_chips"]] & 7)
        assert np.array_equal(h["pos0"] + (w >> 11), oc["sample"][:h["n_chips"]])


def host_decoder_on_every_access_code(dec, ref, ch, al):
    """What the product's host decoder (wm_decoder.c = the reference's state machines, held against them in
    test_host_decoder.py / test_oracle_units.py) does from EVERY access-code chip of a chip stream on:
    {chip index: (consumed, done, packet bytes, L, flags, pkt_rssi, rssi_now, completing sample)}; None if the
    burst runs into the end of the stream."""
    oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al)]
    val, rssi, pos = oc["value"].astype(np.uint32), oc["rssi"].astype(np.uint32), oc["sample"].astype(np.int64)
    db = dec.wm_emu_decoder_bytes()
    dec.wm_decoder_init.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dec.wm_decoder_chip.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
    out = {}
    for i in np.nonzero(val & 2)[0]:
        buf = ctypes.create_string_buffer(db)
        dec.wm_decoder_init(buf, ch)
        st, k, res = 0, 0, None
        for j in range(i, len(val)):
            v = int(val[j])
            if (v & 4) and st == 1:
                res = (k, False); break
            st = dec.wm_decoder_chip(buf, v & 3, int(rssi[j]))
            k += 1
            if st == 2:
                res = (k, True); break
            if st == 0:
                res = (k, False); break
        if res is None:
            out[int(i)] = None
            continue
        d = np.frombuffer(buf.raw, np.uint8)
        # wm_decoder layout (wm_decoder.h): step u16, mode u8, err3of6 u8, c1 u8, frame_b u8, l u16, L u16, [pad], sym u32, mode_bits u32, pkt_rssi u32, packet[292]
        err36, c1, fb = int(d[3]), int(d[4]), int(d[5])
        l_, L = int(d[6]) | int(d[7]) << 8, int(d[8]) | int(d[9]) << 8
        pkt_rssi = int(np.frombuffer(buf.raw[20:24], "<u4")[0])
        out[int(i)] = dict(consumed=res[0], done=res[1], L=L, nbytes=l_, bytes=bytes(d[24:24 + l_]), c1=c1, fb=fb, err36=err36, pkt_rssi=pkt_rssi,
                           rssi_now=int(rssi[i + res[0] - 1]), sample=int(pos[i + res[0] - 1]))
    return out


@pytest.mark.parametrize("seed,amp,kinds,noise", [(1, 60.0, 15, 3.0), (2, 9.0, 15, 3.0), (3, 25.0, 15, 10.0), (4, 60.0, 6, 0.5)])
def test_bursts_inside_the_push_are_decoded_like_the_host_decoder_would(emu_k3, emu_clock, emu_rla, emu_need, oracle, wm, seed, amp, kinds, noise):
    """3-out-of-6 / NRZ / Manchester decode, RSSI gate, framer-reset aborts and the block CRCs on the GPU.
    Every access-code hit whose burst ends inside the push must come back as a WmPkt with exactly the chips-consumed
    count, verdicts, RSSI pair, completing sample and bytes the host decoder produces from the same chips; the rest
    (cut by the end of the push) still arrive as chips."""
    cu8 = wm.synth_capture(seed=5200 + seed, n_samples=1 << 19, kinds=kinds, frames_per_s=150.0, amplitude=amp, noise_sigma=noise)[0]
    if seed == 3:
        cu8[cu8.size // 3 & ~1: cu8.size // 3 + 40000] = 128                # a silent stretch: RSSI gate and framer resets inside bursts
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    fr = framers_on_host(emu_clock, emu_rla, ref)
    rssi = np.zeros((2, fr["Mcap"]), np.uint8)
    for ch in range(2):
        rssi[ch, :ref["m"]] = ref["rssi"][ch].astype(np.uint32).astype(np.uint8)
    hdr, words, pkts, pbytes = bursts_on_host(emu_k3, fr, rssi, decode=True)
    n_done = n_abort = n_raw = 0
    for ch in (0, 1):
        for al in (0, 1):
            want = host_decoder_on_every_access_code(emu_need, ref, ch, al)
            got_p = {int(p["chip0"]): p for p in pkts[(pkts["chain"] == ch) & (pkts["algo"] == al)]}
            got_h = {int(h["chip0"]) for h in hdr[(hdr["chain"] == ch) & (hdr["algo"] == al)]}
            assert sorted(list(got_p) + list(got_h)) == sorted(want), (ch, al)     # every hit exactly once
            for i, w in want.items():
                if i in got_h:
                    n_raw += 1                                   # shipped as chips: the plan reaches past the push (w may even be known)
                    continue
                p = got_p[i]
                assert w is not None, (ch, al, i)
                assert int(p["consumed"]) == w["consumed"] and (p["status"] == 1) == w["done"], (ch, al, i, p, w)
                if w["done"]:
                    fl = int(p["flags"])
                    assert (int(p["L"]), bool(fl & 1), bool(fl & 2), bool(fl & 4)) == (w["L"], bool(w["c1"]), bool(w["fb"]), bool(w["err36"])), (ch, al, i)
                    assert bytes(pbytes[int(p["off"]):int(p["off"]) + w["nbytes"]]) == w["bytes"], (ch, al, i)
                    assert (int(p["pkt_rssi"]), int(p["rssi_now"]), int(p["sample"])) == (w["pkt_rssi"] & 0xFF, w["rssi_now"], w["sample"]), (ch, al, i)
                    n_done += 1
                else:
                    n_abort += 1
    assert n_done > 10 and n_abort > 0 and n_raw <= 8


def _hand_made_stream(ch, vals, rssi_val=40):
    """A chip stream written by hand into the time2 region of chain `ch` (one chip every 8 samples): the framer
    arrays k3 reads, the RSSI rows, and the same chips as an oracle-style record for the host decoder."""
    seg1, seg0 = 32768, 8192
    M = 8 * (len(vals) + 8); Mcap = (M + 255) // 256 * 256
    nseg1, cap1, nseg0, cap0 = 1, seg1 // 4 + 8, 1, seg0 // 2 + 8
    chips1 = np.zeros((2, nseg1, cap1), np.uint32); counts1 = np.zeros((2, nseg1), np.uint32); seen1 = np.zeros((2, nseg1), np.uint32)
    chips0 = np.zeros((2, nseg0, cap0), np.uint32); counts0 = np.zeros((2, nseg0), np.uint32); seen0 = np.zeros((2, nseg0), np.uint32)
    pos = 8 * np.arange(len(vals), dtype=np.uint32) + 16
    chips1[ch, 0, :len(vals)] = (pos << 3) | np.asarray(vals, np.uint32)
    counts1[ch, 0] = len(vals); seen1[ch, 0] = 1
    geo = np.array([M, Mcap, F_T1C1 | F_S1 | F_RLA | F_T2A, 0, seg0, seg1, nseg0, nseg1, cap0, cap1], np.uint64)
    fr = dict(geo=geo, chips=(chips0, chips1), counts=(counts0, counts1), seen=(seen0, seen1), Mcap=Mcap)
    rssi = np.full((2, Mcap), rssi_val, np.uint8)
    rec = np.zeros(len(vals), [("chain", "u1"), ("algo", "u1"), ("sample", "<i8"), ("value", "<u4"), ("rssi", "<u4")])
    rec["chain"], rec["algo"], rec["sample"], rec["value"], rec["rssi"] = ch, 1, pos, vals, rssi_val
    return fr, rssi, dict(chips=rec)


def _s1_chips(data_bytes):
    """access-code chip + Manchester chips of the bytes (01 -> 1, 10 -> 0, s1_packet_decoder.h:35-37) + idle tail."""
    vals = [2]
    for b in data_bytes:
        for k in range(7, -1, -1):
            vals += [0, 1] if (b >> k) & 1 else [1, 0]
    return vals + [0, 1] * 16


@pytest.mark.parametrize("where", ["clean", "last", "interior", "first_data"])
def test_invalid_manchester_pair_anywhere_aborts_the_telegram(emu_k3, emu_need, where):
    """the abort position of the telegram's LAST pair equals the burst length, so it used to read as
    "nothing stopped the decoder" and a phantom line came out (the invalid pair decodes as a 0 bit and may even pass the
    CRC).  The reference resets on it like on any other pair (s1_packet_decoder.h:204-215)."""
    vals = _s1_chips([0x00, 0xA5, 0x3C])                     # L-field 0: the L-field and one CRC pair
    n = 1 + 16 * 3
    if where == "last":
        vals[n - 2:n] = [1, 1]
    elif where == "interior":
        vals[1 + 16 + 6:1 + 16 + 8] = [0, 0]
    elif where == "first_data":
        vals[1 + 16:1 + 16 + 2] = [1, 1]
    fr, rssi, ref = _hand_made_stream(1, vals)
    hdr, words, pkts, pbytes = bursts_on_host(emu_k3, fr, rssi, decode=True)
    want = host_decoder_on_every_access_code(emu_need, ref, 1, 1)[0]
    assert len(hdr) == 0 and len(pkts) == 1
    p = pkts[0]
    assert (int(p["consumed"]), p["status"] == 1) == (want["consumed"], want["done"]), (where, p, want)
    assert want["done"] == (where == "clean")
    if where == "last":
        assert int(p["consumed"]) == n and p["status"] == 2
