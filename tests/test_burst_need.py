"""burst_need (device source, rtl-wmbus_amd/csrc/wm_k3_bursts.h) decides how many chips after an access
code the GPU ships to the host.  Compiled for the host here (tests/emu/need_emu.cpp) and checked against
what the host packet decoder (wm_decoder.c) really consumes: never fewer chips than the decoder wants,
and exactly as many for well-formed headers."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "rtl-wmbus_amd", "csrc")
SO = os.path.join(HERE, "emu", "libneed_emu.so")
SRC = os.path.join(HERE, "emu", "need_emu.cpp")
ENC3OF6 = [0x16, 0x0D, 0x0E, 0x0B, 0x1C, 0x19, 0x1A, 0x13, 0x2C, 0x25, 0x26, 0x23, 0x34, 0x31, 0x32, 0x29]   # EN 13757-4 table


@pytest.fixture(scope="module")
def emu():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("wm_k3_bursts.h", "wm_decoder.c", "wm_decoder.h", "wm_dev.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + CSRC, "-Wno-unknown-pragmas", "-o", SO, SRC,
                        os.path.join(CSRC, "wm_decoder.c")], check=True)
    L = ctypes.CDLL(SO)
    L.wm_emu_burst_need.restype = ctypes.c_uint
    L.wm_emu_burst_need.argtypes = [ctypes.c_uint] * 3
    L.wm_emu_decoder_consumes.restype = ctypes.c_uint
    L.wm_emu_decoder_consumes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint]
    return L


def bits(value, n):
    return [(value >> (n - 1 - i)) & 1 for i in range(n)]


def header(rng, chain):
    """(first chips after the access code, well_formed)"""
    if chain == 1:                                        # S1: Manchester L-field, 01 = one, 10 = zero
        L = int(rng.integers(0, 256))
        pairs = [[0, 1] if b else [1, 0] for b in bits(L, 8)]
        ok = True
        if rng.random() < 0.4:                            # a Manchester violation somewhere in the L-field
            pairs[int(rng.integers(0, 8))] = [int(rng.integers(0, 2))] * 2
            ok = False
        return [c for p in pairs for c in p], ok
    kind = rng.random()
    if kind < 0.45:                                       # T1: L-field as two 3-out-of-6 symbols
        L = int(rng.integers(0, 256))
        return bits(ENC3OF6[L >> 4], 6) + bits(ENC3OF6[L & 15], 6), True
    if kind < 0.8:                                        # C1 frame A / B: 0x54CD / 0x543D, then L (NRZ)
        mode = 0x54C if rng.random() < 0.5 else 0x543
        nib = 0xD if rng.random() < 0.8 else int(rng.integers(0, 16))
        return bits(mode, 12) + bits(nib, 4) + bits(int(rng.integers(0, 256)), 8), nib == 0xD
    return [int(x) for x in rng.integers(0, 2, 24)], False     # noise


@pytest.mark.parametrize("chain", [0, 1])
def test_device_burst_length_covers_what_the_host_decoder_consumes(emu, chain):
    rng = np.random.default_rng(77 + chain)
    limit = 16 * 290 + 64
    exact = total_ok = 0
    for _ in range(int(os.environ.get("WMBUS_NEED_N", "4000"))):
        head, well_formed = header(rng, chain)
        tail = rng.integers(0, 2, limit - len(head))
        if chain == 1 and well_formed:                    # S1 aborts at a Manchester violation: keep the payload legal
            tail = np.repeat(rng.integers(0, 2, (limit - len(head) + 1) // 2), 2)[: limit - len(head)] ^ np.tile([0, 1], limit)[: limit - len(head)]
        chips = np.array(head + [int(x) for x in tail], np.uint8)
        hb = 0
        for b in chips[:24]:
            hb = (hb << 1) | int(b)
        need = emu.wm_emu_burst_need(chain, hb, 24)
        used = emu.wm_emu_decoder_consumes(chain, chips.ctypes.data, chips.size)
        assert used <= need, (chain, hex(hb), used, need)
        if well_formed:
            total_ok += 1
            exact += used == need
    assert total_ok > 100 and exact >= 0.9 * total_ok          # not just an upper bound: the copy is as long as the telegram


def test_short_headers_fall_back_to_the_longest_frame(emu):
    for chain, lim in ((0, 12 * 290 + 1), (1, 16 * 290 + 1)):
        for nb in range(0, 12 if chain == 0 else 16):
            assert emu.wm_emu_burst_need(chain, 0, nb) == lim
