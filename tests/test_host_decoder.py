"""The product's host packet decoders (rtl-wmbus_amd/csrc/wm_decoder.c, inside libwmbus_hip.so)
fed with the oracle's chip log must print the oracle's (= the reference's) lines.  CPU only."""
import ctypes

import numpy as np
import pytest

from cases import SYNTH_CASES, flags_to_oracle_opts, synth_case_capture


class Dec(ctypes.Structure):
    _fields_ = [("step", ctypes.c_uint16), ("mode", ctypes.c_uint8), ("err3of6", ctypes.c_uint8), ("c1", ctypes.c_uint8),
                ("frame_b", ctypes.c_uint8), ("l", ctypes.c_uint16), ("L", ctypes.c_uint16), ("sym", ctypes.c_uint32),
                ("mode_bits", ctypes.c_uint32), ("pkt_rssi", ctypes.c_uint32), ("packet", ctypes.c_uint8 * 292)]


def decode(L, chips, mode, tag):
    d = Dec()
    L.wm_decoder_init(ctypes.byref(d), mode)
    buf = ctypes.create_string_buffer(2048)
    out = []
    for v, r in zip(chips["value"].tolist(), chips["rssi"].tolist()):
        if v & 4:
            d.step = 0
        st = L.wm_decoder_chip(ctypes.byref(d), v & 3, r)
        if st == 2:
            n = L.wm_decoder_format(ctypes.byref(d), tag, b"TS", r, buf, 2048, None)
            out.append(buf.raw[:n].decode())
    return out


@pytest.mark.parametrize("case", [SYNTH_CASES[k] for k in (1, 2, 9)], ids=lambda c: c["id"])
def test_host_decoder_equals_oracle_lines(wm, oracle, case):
    L = wm.lib()
    L.wm_decoder_format.restype = ctypes.c_size_t
    L.wm_decoder_format.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint, ctypes.c_char_p,
                                    ctypes.c_size_t, ctypes.c_void_p]
    cu8, _ = synth_case_capture(wm, case)
    r = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), chips=True)
    lines = r["text"].splitlines(keepends=True)
    total = 0
    for ch, modes in ((0, ("T1", "C1")), (1, ("S1",))):
        for al, tag in ((0, b"rla;"), (1, b"t2a;")):
            c = r["chips"][(r["chips"]["chain"] == ch) & (r["chips"]["algo"] == al)]
            got = decode(L, c, ch, tag)
            want = [l for l in lines if l.startswith(tag.decode()) and l.split(";")[1] in modes]
            assert got == want
            total += len(got)
    assert total == len(lines)


def test_crc16_known_answer(wm):
    L = wm.lib()
    L.wm_crc16.restype = ctypes.c_uint16
    L.wm_crc16.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    # first block of the samples2 telegram: CRC-16/EN-13757 check value
    assert L.wm_crc16(b"123456789", 9) == 0xC2B7


def test_twin_filter_drops_the_second_of_a_pair(wm):
    """cfg.dedup_twins / CLI -U (README.md:105-108): the later of two lines with the same payload from
    different framers within one telegram time goes; different payloads, the same framer, or a long gap stay."""
    import ctypes, json, os
    from conftest import GOLDEN
    L = wm.lib()

    class Twin(ctypes.Structure):
        _fields_ = [("sample", ctypes.c_uint64), ("hash", ctypes.c_uint64), ("algo", ctypes.c_uint8), ("valid", ctypes.c_uint8)]
    L.wm_twin_check.argtypes = [ctypes.POINTER(Twin), ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t]
    lines = json.load(open(os.path.join(GOLDEN, "bundled.json")))["rtlsdr_868.950M_1M6_samples2.cu8|-v"].splitlines(True)
    assert len(lines) == 4 and lines[0].split(";")[-1] == lines[1].split(";")[-1] and lines[2].split(";")[-1] != lines[3].split(";")[-1]
    st = (Twin * 2)()
    drop = [L.wm_twin_check(st, 0, 1 if ln.startswith("t2a") else 0, 1000 + 10 * k, ln.encode(), len(ln)) for k, ln in enumerate(lines)]
    assert drop == [0, 1, 0, 0]                               # the 71200023 twin goes; the 64700082 pair differs in its last bytes
    st = (Twin * 2)()
    a, b = lines[0].encode(), lines[1].encode()
    assert L.wm_twin_check(st, 0, 1, 10, a, len(a)) == 0
    assert L.wm_twin_check(st, 0, 1, 20, a, len(a)) == 0      # the same framer again: a repeated transmission, kept
    assert L.wm_twin_check(st, 0, 0, 10 + 12 * 290 * 8 + 5000, b, len(b)) == 0     # the other framer, but a telegram time later: kept
    assert L.wm_twin_check(st, 0, 1, 10 + 12 * 290 * 8 + 5100, a, len(a)) == 1     # ... and ITS twin goes
