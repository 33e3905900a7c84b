"""Stage-level pinning of the oracle against the reference's own functions (via oracle/ref_probe.c,
which includes the reference translation unit).  Skipped where /root/reference was never built."""
import os
import subprocess

import numpy as np
import pytest

import oracle_ffi as O
from cases import flags_to_oracle_opts

pytestmark = pytest.mark.skipif(not os.path.exists(O.REF_PROBE), reason="oracle/_ref/ref_probe not built")


def test_decoder_tables_match_reference():
    out = subprocess.run([O.REF_PROBE, "tables"], capture_output=True, text=True, check=True).stdout
    tab = {l.split()[0]: list(map(int, l.split()[1:])) for l in out.splitlines()}
    sym = [0x16, 0x0D, 0x0E, 0x0B, 0x1C, 0x19, 0x1A, 0x13, 0x2C, 0x25, 0x26, 0x23, 0x34, 0x31, 0x32, 0x29]
    nib = [255] * 64
    for n, s in enumerate(sym):
        nib[s] = n
    assert tab["LO"] == nib
    assert tab["HI"] == [255 if v == 255 else v << 4 for v in nib]
    assert tab["LEN"] == [1 + L + 2 * (1 + ((L - 9 + 15) // 16 if L > 9 else 0)) for L in range(256)]
    crc = []
    for v in range(256):
        r = v << 8
        for _ in range(8):
            r = ((r << 1) ^ 0x3D65) & 0xFFFF if r & 0x8000 else (r << 1) & 0xFFFF
        crc.append(r)
    assert tab["CRC"] == crc
    assert tab["DEGLITCH_T"][:64] == [int(bin(i).count("1") >= 3) for i in range(64)]
    assert tab["DEGLITCH_S"] == [(0xFEEA >> i) & 1 for i in range(16)]


@pytest.mark.parametrize("inaccurate", [False, True])
def test_soft_symbol_taps_bit_identical(wm, tmp_path, inaccurate):
    cu8, _ = wm.synth_capture(seed=7, n_samples=1 << 18, kinds=15, frames_per_s=80.0, amplitude=30.0)
    prefix = str(tmp_path / "taps")
    subprocess.run([O.REF_PROBE, "stages", prefix] + (["a"] if inaccurate else []), input=cu8.tobytes(), check=True)
    r = O.run(cu8, flags_to_oracle_opts(O, ["-a"] if inaccurate else []), taps=True)
    for ch in (0, 1):
        for name, key in (("iq", "iq"), ("draw", "dphi_raw"), ("dphi", "dphi"), ("rssi", "rssi"), ("clk", "clk")):
            ref = np.fromfile(f"{prefix}.{name}{ch}.f32", np.float32)
            assert np.array_equal(ref.view(np.uint32), r[key][ch].view(np.uint32)), (name, ch)


@pytest.mark.parametrize("inaccurate", [False, True])
def test_polyphase_prefilter_taps_bit_identical(wm, tmp_path, inaccurate):
    """the polyphase low-pass the reference defines (ppf.h:46-59, rtl_wmbus.c:258-294)
    but never calls.  The oracle's optional prefilter is pinned on the reference's own function:
    filtered (i,q), discriminator, FIR, RSSI and clock-filter taps of the reference stage functions
    fed by lp_ppf_butter_1600kHz_160kHz_200kHz equal the oracle's with prefilter=1, bit for bit."""
    cu8, _ = wm.synth_capture(seed=17, n_samples=1 << 18, kinds=15, frames_per_s=80.0, amplitude=30.0)
    prefix = str(tmp_path / "ppf")
    subprocess.run([O.REF_PROBE, "stages", prefix, "pa" if inaccurate else "p"], input=cu8.tobytes(), check=True)
    r = O.run(cu8, O.make_opts(accurate_atan=0 if inaccurate else 1, prefilter=1), taps=True)
    for ch in (0, 1):
        for name, key in (("iq", "iq"), ("draw", "dphi_raw"), ("dphi", "dphi"), ("rssi", "rssi"), ("clk", "clk")):
            ref = np.fromfile(f"{prefix}.{name}{ch}.f32", np.float32)
            assert np.array_equal(ref.view(np.uint32), r[key][ch].view(np.uint32)), (name, ch)
    assert len(r["text"].splitlines()) >= 4      # the synthetic telegrams still decode behind this filter


def test_reference_decoder_accepts_oracle_chip_log(wm):
    """Feed the oracle's chip log to the reference's packet decoders: same lines as the oracle."""
    cu8, _ = wm.synth_capture(seed=9, n_samples=1 << 19, kinds=15, frames_per_s=80.0, amplitude=30.0)
    r = O.run(cu8, flags_to_oracle_opts(O, ["-v"]), chips=True)
    lines = r["text"].splitlines()
    for ch, mode in ((0, ("T1", "C1")), (1, ("S1",))):
        for al, tag in ((0, "rla;"), (1, "t2a;")):
            c = r["chips"][(r["chips"]["chain"] == ch) & (r["chips"]["algo"] == al)]
            raw = np.stack([c["value"], c["rssi"]], axis=1).astype(np.uint8).tobytes()
            out = subprocess.run([O.REF_PROBE, "chips", str(ch), tag], input=raw, capture_output=True, check=True).stdout
            got = O.mask_ts(out).decode().splitlines()
            want = [l for l in lines if l.startswith(tag) and l.split(";")[1] in mode]
            assert got == want


@pytest.mark.parametrize("which", [1, 2])
def test_atan2_approximations_equal_the_references_own_functions(which):
    """atan2.h:14-74: the two approximations the reference keeps behind `#elif 0` / `#else` -- the oracle's
    restatement (wmo_opts.atan_mode) against the reference's functions, bit for bit, on discriminator-like
    operands (products of small integers), special values and random floats."""
    import ctypes
    rng = np.random.default_rng(which)
    ints = rng.integers(-4000, 4001, (200000, 2)).astype(np.float32)
    scaled = (ints * np.float32(1 / 64)).astype(np.float32)
    rnd = rng.standard_normal((100000, 2)).astype(np.float32) * np.float32(10.0) ** rng.integers(-12, 6, (100000, 1)).astype(np.float32)
    special = np.array([[0, 0], [0, 1], [0, -1], [1, 0], [-1, 0], [1e-10, 1], [-1e-10, 1], [1e-11, -1], [3, 3], [3, -3], [-3, 3], [-3, -3]], np.float32)
    pairs = np.concatenate([ints, scaled, rnd, special]).astype(np.float32)            # columns: imaginary, real
    p = subprocess.run([O.REF_PROBE, "atan", str(which)], input=pairs.tobytes(), stdout=subprocess.PIPE, check=True)
    want = np.frombuffer(p.stdout, np.float32)
    got = np.zeros(len(pairs), np.float32)
    L = O.lib()
    L.wmo_atan2_approx.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    im, re = np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])
    L.wmo_atan2_approx(which, im.ctypes.data, re.ctypes.data, got.ctypes.data, len(pairs))
    assert len(want) == len(got) and np.array_equal(want.view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("which", [0, 1])
def test_low_pass_filters_equal_the_references_own_functions_signed_zeros_included(which):
    """rtl_wmbus.c:369-391 over fir.h:48-72: the oracle's FIR against the reference's own functions on rows a capture hardly ever
    produces.  The sum starts from +0, so an output is never -0 -- the property the clock kernel's slicer (sign bit instead of
    `>= 0`, rtl_wmbus.c:1059) rests on and the device self-test checks on the same rows (test_gpu_parity.py)."""
    for seed in (1, 2):
        for x in O.fir_rows(seed):
            p = subprocess.run([O.REF_PROBE, "fir", str(which)], input=x.tobytes(), stdout=subprocess.PIPE, check=True)
            want = np.frombuffer(p.stdout, np.float32)
            got = O.fir(which, x)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
            assert not np.any(got.view(np.uint32) == 0x80000000), "a soft symbol is never -0"
