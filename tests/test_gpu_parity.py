"""Parity tests proper: the HIP path (through the C ABI) against the committed golden output of
the reference and against the oracle -- datagram text byte-for-byte, soft symbols / RSSI / slicer
bits / chip streams bit-for-bit (tolerance 0: every float op is rounded like the reference's)."""
import json
import os
import subprocess

import numpy as np
import pytest

from cases import BUNDLED_CASES, SYNTH_CASES, flags_to_kwargs, flags_to_oracle_opts, synth_case_capture
from conftest import GOLDEN, SAMPLES
from cases import S2 as S2_NAME

pytestmark = pytest.mark.gpu

BUNDLED = json.load(open(os.path.join(GOLDEN, "bundled.json")))
SYNTH = json.load(open(os.path.join(GOLDEN, "synthetic.json")))
SOFT_SYMBOL_TOLERANCE = 0.0   # |delta_phi_gpu - delta_phi_reference|; the path is bit-exact


def compare_taps(rx, ref, stream=0, chains=(0, 1)):
    m = ref["m"]
    for ch in chains:
        d = rx.read_tap("dphi", ch, stream, m)       # FIR output (the DC remover of -o runs in the clock kernel)
        assert np.abs(d - ref["dphi_fir"][ch]).max(initial=0.0) <= SOFT_SYMBOL_TOLERANCE
        assert np.array_equal(d.view(np.uint32), ref["dphi_fir"][ch].view(np.uint32))
        assert np.array_equal(rx.read_tap("rssi", ch, stream, m), ref["rssi"][ch].astype(np.uint32).astype(np.uint8))
        assert np.array_equal(rx.read_tap("bits", ch, stream, m), ref["bit"][ch])


def compare_chips(rx, ref, stream=0, chains=(0, 1), algos=(0, 1)):
    for ch in chains:
        for al in algos:
            w, pos = rx.read_chips(ch, al, stream)
            oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al)]
            assert len(w) == len(oc), (ch, al)
            assert np.array_equal(w & 0xFF, oc["value"]) and np.array_equal((w >> 8) & 0xFF, oc["rssi"])
            assert np.array_equal(pos, oc["sample"])


def test_device_divide_on_the_adversarial_operands(wm):
    """wm_div_dom (wm_exact.h) makes ONE Markstein correction q + (a - b q) y from q = RN(a y), y = RN(1/b) -- exact if q is a faithful
    quotient, and the analytic bound for q is about 1.5 ulp where the quotient's significand is close to 2 and y was rounded UP by
    nearly half an ulp.  So those operands are tested by construction, not by sampling: for EVERY significand of b the
    dividends just below 2 b and just above b, and for the 30 000 significands whose reciprocal rounds worst, 64 dividends a side
    more plus the quotients next to a rounding boundary -- against the host's IEEE division (numpy float32 `/`)."""
    def check(a, b, what):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        r = wm.selftest_math(a, b)
        want = (a / b).astype(np.float32)
        bad = np.flatnonzero(r["div"].view(np.uint32) != want.view(np.uint32))
        assert bad.size == 0, (what, bad.size, a[bad[:4]], b[bad[:4]], r["div"][bad[:4]], want[bad[:4]])

    worst = []
    for slab in range(2):                                     # all 2^23 significands: b = (2^23 + k), an integer below 2^24 like the kernels' operands
        k = np.arange(slab << 22, (slab + 1) << 22, dtype=np.int64)
        b = (k + (1 << 23)).astype(np.float32)
        two_b = (2 * (k + (1 << 23))).astype(np.float32)
        for steps in (1, 2, 3):
            below = (two_b.view(np.uint32) - np.uint32(steps)).view(np.float32)      # quotient = 2 - a few ulps
            check(below, b, f"just below 2b, slab {slab}")
            above = (b.view(np.uint32) + np.uint32(steps)).view(np.float32)          # quotient = 1 + a few ulps
            check(above, b, f"just above b, slab {slab}")
        y64 = 1.0 / b.astype(np.float64)
        y32 = y64.astype(np.float32)
        err = (y32.astype(np.float64) - y64) / np.spacing(y32).astype(np.float64)     # in ulps of y: +0.5 = rounded up by half an ulp
        idx = np.argsort(-np.abs(err))[:15000]
        worst.append(b[idx])
    b = np.concatenate(worst)
    rng = np.random.default_rng(5)
    for steps in range(1, 65):
        check(((2 * b).view(np.uint32) - np.uint32(steps)).view(np.float32), b, f"worst reciprocals, 2b - {steps} ulp")
        check((b.view(np.uint32) + np.uint32(steps)).view(np.float32), b, f"worst reciprocals, b + {steps} ulp")
    for _ in range(16):                                       # quotients next to a rounding boundary: a = RN(b (q + half an ulp of q)) for q near 2
        q = (np.float32(2.0).view(np.uint32) - rng.integers(1, 1 << 12, b.size).astype(np.uint32)).view(np.float32)
        a = (b.astype(np.float64) * (q.astype(np.float64) + 0.5 * np.spacing(q).astype(np.float64))).astype(np.float32)
        check(a, b, "next to a rounding boundary")
        check((a.view(np.uint32) + np.uint32(1)).view(np.float32), b, "next to a rounding boundary + 1")
        check((a.view(np.uint32) - np.uint32(1)).view(np.float32), b, "next to a rounding boundary - 1")


def test_device_arithmetic_is_ieee_and_glibc_exact(wm, oracle, libm_is_glibc_235):
    """The kernels' scalar arithmetic (wm_exact.h) on the device against this host's IEEE / glibc:
      * square root: EXHAUSTIVE over the RSSI operand domain, every integer in [0, 2^24) (the
        kernels take the root of the unscaled integer sums i^2 + q^2) plus the k/64, k/256 forms;
      * divide: 5*10^7 operand pairs of the discriminator's domain (integers below 2^24 and the
        range-reduced quotients of the arctangent);
      * atan2f and the whole discriminator: 3*10^7 pairs, every one compared with libm's atan2f
        (the reference's atan2.h:7-10 expression)."""
    L = oracle.lib()
    import ctypes
    fp = ctypes.POINTER(ctypes.c_float)
    for f in (L.wmo_libm_atan2f, L.wmo_ieee_div):
        f.argtypes = [fp, fp, fp, ctypes.c_size_t]; f.restype = None
    L.wmo_ieee_sqrt.argtypes = [fp, fp, ctypes.c_size_t]; L.wmo_ieee_sqrt.restype = None

    def host2(fn, x, y):
        out = np.empty_like(x); fn(x.ctypes.data_as(fp), y.ctypes.data_as(fp), out.ctypes.data_as(fp), x.size); return out

    # ---- glibc 2.35's known answers first: the device must reproduce them whatever libm this host has
    kat = np.fromfile(os.path.join(GOLDEN, "atan2f_kat.bin"), "<u4").reshape(-1, 3)
    r = wm.selftest_math(kat[:, 0].copy().view(np.float32), kat[:, 1].copy().view(np.float32))
    assert np.array_equal(r["atan2"].view(np.uint32), kat[:, 2]), "device atan2f against glibc 2.35's known answers"
    rng = np.random.default_rng(11)
    # ---- sqrt, exhaustive on the integer domain, in 4 slabs of 2^22
    for slab in range(4):
        a = np.arange(slab << 22, (slab + 1) << 22, dtype=np.float32)
        r = wm.selftest_math(a, np.ones_like(a))
        want = np.empty_like(a); L.wmo_ieee_sqrt(a.ctypes.data_as(fp), want.ctypes.data_as(fp), a.size)
        assert np.array_equal(r["sqrt"].view(np.uint32), want.view(np.uint32)), f"sqrt slab {slab}"
    a = np.concatenate([np.arange(0, 2 * 1016 * 1016 + 1) / 64.0, rng.integers(0, 2 * 2880 * 2880, 1 << 21) / 256.0]).astype(np.float32)
    r = wm.selftest_math(a, np.ones_like(a))
    assert np.array_equal(r["sqrt"].view(np.uint32), np.sqrt(a).view(np.uint32))

    # ---- divide / atan2 / discriminator on discriminator-shaped operands
    n = 1 << 22
    for rnd in range(12):
        lim, sc = ((1016, 8.0), (2880, 16.0))[rnd & 1]
        v = rng.integers(-lim, lim + 1, (4, n))
        weak = rng.random((4, n)) < 0.3
        v = np.where(weak, rng.integers(-12, 13, (4, n)), v)
        if rnd >= 8:                                              # unscaled boxcar sums, as the kernels feed them
            sc = 1.0
        i, q, ip, qp = [(v[k] / sc).astype(np.float32) for k in range(4)]
        re = (i * ip - q * (-qp)).astype(np.float32); im = (i * (-qp) + q * ip).astype(np.float32)
        r = wm.selftest_math(im, re)
        nz = re != 0
        want = host2(L.wmo_ieee_div, im, np.where(nz, re, np.float32(1)))
        assert np.array_equal(r["div"][nz].view(np.uint32), want[nz].view(np.uint32)), f"divide round {rnd}"
        if not libm_is_glibc_235:
            continue                                              # this host's libm is another generation: nothing to compare the arctangent with
        want = host2(L.wmo_libm_atan2f, im, re)
        assert np.array_equal(r["atan2"].view(np.uint32), want.view(np.uint32)), f"atan2f round {rnd}"
        # range-reduced operands of the second division: t in [7/16, 39/16) -> (2t-1)/(2+t), (t-1)/(t+1), (t-1.5)/(1+1.5t); -1/t
        t = np.abs(want) .astype(np.float32) + np.float32(0.4375)
        num = np.where(t < 0.6875, 2 * t - 1, np.where(t < 1.1875, t - 1, t - np.float32(1.5))).astype(np.float32)
        den = np.where(t < 0.6875, 2 + t, np.where(t < 1.1875, t + 1, 1 + np.float32(1.5) * t)).astype(np.float32)
        r2 = wm.selftest_math(num, den)
        assert np.array_equal(r2["div"].view(np.uint32), host2(L.wmo_ieee_div, num, den).view(np.uint32)), f"reduced divide round {rnd}"
        # whole discriminator: o_disc[j] = discriminator(i=a[j], q=b[j], i'=b[j+1], q'=a[j+1])
        r3 = wm.selftest_math(i, q)
        ipp, qpp = np.roll(q, -1), np.roll(i, -1)
        re3 = (i * ipp - q * (-qpp)).astype(np.float32); im3 = (i * (-qpp) + q * ipp).astype(np.float32)
        want3 = (host2(L.wmo_libm_atan2f, im3, re3) * np.float32(0.3183098861837907)).astype(np.float32)
        assert np.array_equal(r3["disc"].view(np.uint32), want3.view(np.uint32)), f"discriminator round {rnd}"
    # signed zeros and axes
    z = np.array([0.0, -0.0, 3.5, -3.5, 0.0, -0.0, 2.25, -2.25], np.float32)
    yy, xx = np.meshgrid(z, z)
    yy = np.ascontiguousarray(yy.ravel()); xx = np.ascontiguousarray(xx.ravel())
    r = wm.selftest_math(yy, xx)
    if not libm_is_glibc_235:
        pytest.skip("device arithmetic reproduces glibc 2.35's known answers; this host's libm is another generation (the reference itself "
                    "would print other soft symbols here), so the libm comparisons were left out")
    assert np.array_equal(r["atan2"].view(np.uint32), host2(L.wmo_libm_atan2f, yy, xx).view(np.uint32))


def test_device_low_pass_filters_signed_zeros_included(wm, oracle):
    """The demodulation kernel's two low-pass filters (the functions its stage B inlines, run by wmbus_selftest_fir) against the
    oracle's FIR on rows a capture hardly ever produces -- signed zeros above all: the reference's sum starts from +0
    (fir.h:56), so a soft symbol is never -0, and the clock kernel takes the slicer's bit (`>= 0`, rtl_wmbus.c:1059) from the
    sign bit.  (Round 5: the compiler had turned 0 + b0 x, b0 < 0, into a negated operand of the next subtraction --
    -0 where the reference has +0 under the pattern of tests/oracle_ffi.py fir_rows; the first tap is an fma with +0 now.)"""
    for seed in (1, 2, 3):
        for x in oracle.fir_rows(seed):
            y11, y46 = wm.selftest_fir(x)
            for which, got in ((0, y11), (1, y46)):
                want = oracle.fir(which, x)[48:]
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, which)
                assert not np.any(got.view(np.uint32) == 0x80000000)


@pytest.mark.parametrize("name,flags", BUNDLED_CASES, ids=[f"{n[14:22]}:{' '.join(f)}" for n, f in BUNDLED_CASES])
def test_bundled_captures_match_reference_golden(wm, name, flags):
    cu8 = np.fromfile(os.path.join(SAMPLES, name), np.uint8)
    with wm.Receiver(n_streams=1, max_push_bytes=4 << 20, **flags_to_kwargs(flags)) as rx:
        assert rx.run(cu8)[0] == BUNDLED[f"{name}|{' '.join(flags)}"]


@pytest.mark.parametrize("case", SYNTH_CASES, ids=[c["id"] for c in SYNTH_CASES])
def test_synthetic_captures_match_reference_golden_and_oracle(wm, oracle, case):
    cu8, _ = synth_case_capture(wm, case)
    kw = flags_to_kwargs(case["flags"])
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, case["flags"]), taps=True, chips=True)
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, **kw) as rx:
        text = rx.run(cu8)[0]
        chains = [c for c, on in ((0, kw.get("t1c1", True)), (1, kw.get("s1", True))) if on]
        algos = [a for a, on in ((0, kw.get("rla", True)), (1, kw.get("time2", True))) if on]
        compare_taps(rx, ref, chains=chains)
        compare_chips(rx, ref, chains=chains, algos=algos)
    assert text == SYNTH[case["id"]]
    assert text == ref["text"]


@pytest.mark.parametrize("flags", [[], ["-a"], ["-o"], ["-p", "S"]], ids=lambda f: " ".join(f) or "default")
def test_polyphase_prefilter_matches_oracle(wm, oracle, samples, flags):
    """cfg.prefilter = POLYPHASE (ppf.h:46-59 as rtl_wmbus.c:258-294 drives it).  The
    oracle's polyphase stage is pinned on the reference's own function (test_oracle_units.py); here
    the HIP kernel must reproduce the oracle: soft symbols, RSSI, slicer bits, every chip, the text."""
    kw = flags_to_kwargs(flags + ["-v"])
    oo = flags_to_oracle_opts(oracle, flags + ["-v"]); oo.prefilter = 1
    chains = [c for c, on in ((0, kw.get("t1c1", True)), (1, kw.get("s1", True))) if on]
    for cu8 in (samples["samples2"],
                wm.synth_capture(seed=41, n_samples=1 << 19, kinds=15, frames_per_s=90.0, amplitude=35.0)[0]):
        ref = oracle.run(cu8, oo, taps=True, chips=True)
        with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, prefilter=1, **kw) as rx:
            text = rx.run(cu8)[0]
            compare_taps(rx, ref, chains=chains)
            compare_chips(rx, ref, chains=chains)
        assert text == ref["text"]
    assert len(text.splitlines()) >= 4


def test_polyphase_prefilter_streams_and_rejects_other_rates(wm, oracle):
    cu8, _ = wm.synth_capture(seed=43, n_samples=1 << 18, kinds=15, frames_per_s=90.0, amplitude=35.0)
    oo = oracle.make_opts(prefilter=1)
    ref = oracle.run(cu8, oo)["text"]
    with wm.Receiver(n_streams=1, max_push_bytes=1 << 18, prefilter=1) as rx:
        assert rx.run(cu8, push_bytes=4096 * 9)[0] == ref      # filter history carried across pushes
    with pytest.raises(wm.WmbusError):
        wm.Receiver(n_streams=1, decimation=3, prefilter=1)
    with pytest.raises(wm.WmbusError):
        wm.Receiver(n_streams=1, simultaneous=True, prefilter=1)


@pytest.mark.parametrize("seg_len,w0,w1,lb", [(4096, 1024, 1024, 64), (8192, 4096, 8192, 256), (65536, 24576, 49152, 1024),
                                               (262144, 12288, 24576, 1024)])
def test_result_independent_of_segmentation(wm, oracle, samples, seg_len, w0, w1, lb):
    """Short warm-ups force hand-off verification failures: the re-run path must restore exactness."""
    cu8 = samples["samples2"]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, seg_len=seg_len, rla_seg_len=min(65536, max(1024, seg_len // 4)),
                     warmup_t1c1=w0, warmup_s1=w1, rla_lookback=lb) as rx:
        text = rx.run(cu8)[0]
        tim = rx.timing()
        compare_chips(rx, ref)
    assert text == ref["text"]
    if seg_len == 4096:
        assert tim["clock_reruns"] > 0 and tim["rla_reruns"] > 0   # the slow path really ran


@pytest.mark.parametrize("push_bytes", [4096, 8192 * 5, 1 << 18, 1 << 20])
def test_streaming_pushes_equal_one_shot(wm, oracle, push_bytes):
    """State (filters, framers, decoders mid-telegram) is carried across pushes exactly."""
    cu8, _ = wm.synth_capture(seed=77, n_samples=(1 << 19) if push_bytes < 65536 else (1 << 20), kinds=15,
                              frames_per_s=80.0, amplitude=40.0)
    if push_bytes == 4096:
        cu8 = cu8[: 1 << 18]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]))["text"]
    with wm.Receiver(n_streams=1, max_push_bytes=1 << 20) as rx:
        assert rx.run(cu8, push_bytes=push_bytes)[0] == ref


def test_decimation_phase_carries_across_pushes(wm, oracle):
    cu8, _ = wm.synth_capture(seed=78, n_samples=1 << 19, fs_khz=2400, kinds=15, frames_per_s=80.0)
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-d", "3", "-v"]))["text"]
    with wm.Receiver(n_streams=1, max_push_bytes=1 << 20, decimation=3) as rx:
        assert rx.run(cu8, push_bytes=4096 * 7)[0] == ref     # 2048*7 samples: not a multiple of 3


def test_many_streams_in_one_batch(wm, oracle):
    n_streams, n = 48, 1 << 18
    caps = [wm.synth_capture(seed=900 + s, n_samples=n, kinds=15, frames_per_s=100.0, amplitude=[60, 25, 9][s % 3])[0]
            for s in range(n_streams)]
    with wm.Receiver(n_streams=n_streams, max_push_bytes=2 * n, seg_len=16384, warmup_t1c1=8192, warmup_s1=16384) as rx:
        texts = rx.run(caps)
        for s in (0, 17, 47):
            ref = oracle.run(caps[s], flags_to_oracle_opts(oracle, ["-v"]), taps=True, chips=True)
            compare_taps(rx, ref, stream=s)
            compare_chips(rx, ref, stream=s)
    for s in range(n_streams):
        assert texts[s] == oracle.run(caps[s], flags_to_oracle_opts(oracle, ["-v"]))["text"], s


@pytest.mark.parametrize("flags", [["-v"], ["-o", "-v"]], ids=lambda f: " ".join(f))
def test_wave_of_64_captures_takes_the_cooperative_path(wm, oracle, flags):
    """n_streams a multiple of 64: the clock kernel's wave = 64 consecutive captures of one (chain,
    segment) and fetches their rows cooperatively through LDS (bench.py's configuration).  Every
    capture's text, chips and soft symbols against the oracle; several segments per capture, a
    second push (carried state) and different signal levels per capture."""
    n_streams, n = 128, 1 << 18
    caps = [wm.synth_capture(seed=1300 + s, n_samples=n, kinds=15, frames_per_s=100.0, amplitude=[60, 25, 9, 40][s % 4])[0]
            for s in range(n_streams)]
    kw = flags_to_kwargs(flags)
    with wm.Receiver(n_streams=n_streams, max_push_bytes=n, **kw) as rx:      # two pushes of n bytes each
        texts = rx.run(caps)
        tim = rx.timing()
        for s in (0, 63, 64, 127):                                            # taps/chips of the LAST push
            ref = oracle.run(caps[s], flags_to_oracle_opts(oracle, flags), taps=True, chips=True)
            m_half = ref["m"] // 2
            d = rx.read_tap("dphi", 0, s, m_half)
            assert np.array_equal(d.view(np.uint32), ref["dphi_fir"][0][m_half:].view(np.uint32))
            for ch in (0, 1):
                for al in (0, 1):
                    w, pos = rx.read_chips(ch, al, s)
                    oc = ref["chips"][(ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al) & (ref["chips"]["sample"] >= m_half)]
                    assert len(w) == len(oc) and np.array_equal(w & 0xFF, oc["value"]) and np.array_equal((w >> 8) & 0xFF, oc["rssi"])
                    assert np.array_equal(pos, oc["sample"])
    for s in range(n_streams):
        assert texts[s] == oracle.run(caps[s], flags_to_oracle_opts(oracle, flags))["text"], s
    assert sum(len(t.splitlines()) for t in texts) > 4 * n_streams


def test_edge_inputs(wm, oracle):
    rng = np.random.default_rng(5)
    cases = {
        "all_zero_bytes": np.zeros(1 << 17, np.uint8),
        "all_128": np.full(1 << 17, 128, np.uint8),
        "full_scale_noise": rng.integers(0, 256, 1 << 18, dtype=np.uint8),
        "square_wave": np.tile(np.array([255, 0, 0, 255], np.uint8), 1 << 16),
        "one_block": rng.integers(100, 156, 4096, dtype=np.uint8),
    }
    for name, cu8 in cases.items():
        for flags in (["-v"], ["-s", "-o", "-v"]):
            ref = oracle.run(cu8, flags_to_oracle_opts(oracle, flags), taps=True, chips=True)
            with wm.Receiver(n_streams=1, max_push_bytes=1 << 18, **flags_to_kwargs(flags)) as rx:
                assert rx.run(cu8)[0] == ref["text"], (name, flags)
                compare_taps(rx, ref)
                compare_chips(rx, ref)


@pytest.mark.parametrize("flags", [["-v"], ["-s", "-v"]], ids=lambda f: " ".join(f))
def test_signal_followed_by_exact_silence_repairs_the_rssi_filter(wm, oracle, flags):
    """Exact-zero input after a signal: the RSSI EMA decays through ~90 more samples (down to the
    subnormals) while a 48-sample warm-up from zero is already at zero, so the tile hand-off cannot be
    certified.  Such tiles are re-run sequentially from the predecessor's exact state: same bytes as
    the oracle, and the repair path is seen to have run."""
    sig, _ = wm.synth_capture(seed=77, n_samples=1 << 17, kinds=15, frames_per_s=120.0, amplitude=50.0)
    parts = [sig[: 37 * 4096], np.full(21 * 4096 + 2 * 977 * 2, 128, np.uint8), sig[37 * 4096: 90 * 4096],
             np.full(64 * 4096, 127, np.uint8), sig[90 * 4096:]]
    cu8 = np.concatenate(parts)
    cu8 = cu8[: cu8.size // 4096 * 4096]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, flags), taps=True, chips=True)
    kw = flags_to_kwargs(flags)
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, **kw) as rx:
        assert rx.run(cu8)[0] == ref["text"]
        assert rx.timing()["ema_retries"] > 0
        compare_taps(rx, ref)
        compare_chips(rx, ref)
    with wm.Receiver(n_streams=1, max_push_bytes=1 << 16, **kw) as rx:          # and across push boundaries
        assert rx.run(cu8, push_bytes=1 << 16)[0] == ref["text"]


@pytest.mark.parametrize("flags", [["-v"], ["-s", "-v"]], ids=lambda f: " ".join(f))
def test_rssi_on_demand_falls_back_to_the_full_pass_where_it_cannot_prove_a_value(wm, oracle, flags):
    """Contexts without debug views compute the RSSI only for the tiles bursts touch, every lane proving its start by a
    bracket of two trajectories (wm_k1_demod.h).  Telegrams right behind exact silence / constant input sit on tiles whose
    brackets stay open: the push is finished by the full pass (slow path), and the text is the oracle's either way."""
    centres = dict(t1c1_center_khz=325.0, s1_center_khz=-325.0) if "-s" in flags else {}
    sig, _ = wm.synth_capture(seed=77, n_samples=1 << 20, kinds=15, frames_per_s=400.0, amplitude=50.0, noise_sigma=0.0, **centres)
    quiet = [np.full(21 * 4096 + 2 * 977 * 2, 128, np.uint8), np.full(64 * 4096, 127, np.uint8), np.tile(np.array([127, 128], np.uint8), 16 * 2048)]
    parts = []
    for k in range(12):                                      # signal and silence alternate; without noise the gaps between telegrams are constant input too
        parts += [sig[k * 40 * 4096 + 2 * 131 * k: (k + 1) * 40 * 4096 + 2 * 131 * k], quiet[k % 3]]
    cu8 = np.concatenate(parts)
    cu8 = cu8[: cu8.size // 4096 * 4096]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, flags))
    assert len(ref["text"].splitlines()) > 10
    kw = flags_to_kwargs(flags)
    slow = 0
    for push in (cu8.size, 1 << 16, 4096 * 5):
        with wm.Receiver(n_streams=1, max_push_bytes=push, keep_taps=False, **kw) as rx:
            text = []
            for off in range(0, cu8.size, push):
                rx.push([cu8[off:off + push]])
                text += [ln["text"] for ln in rx.lines()]
                tm = rx.timing()
                slow += tm["slow_path"]
                # how the push got its RSSI is reported: a full pass behind an unprovable value is FELL_BACK (and slow)
                assert tm["rssi_mode"] != wm.RSSI_EVERY_SAMPLE and (tm["rssi_mode"] != wm.RSSI_FELL_BACK or tm["slow_path"] == 1)
            assert "".join(text) == ref["text"], push
    assert slow > 0                                         # the full pass was needed somewhere


def test_very_long_exact_silence_does_not_overflow_the_chip_regions(wm, oracle):
    """A run of identical chips as long as the silence before it ends at one edge (the reference's
    loop emits them all: 131 072 chips after a million silent samples).  The kernel materialises
    8192 per edge -- more than any decoder consumes after an access code -- and counts the rest;
    datagrams and RSSI are unaffected and nothing overflows."""
    sig, _ = wm.synth_capture(seed=91, n_samples=1 << 18, kinds=15, frames_per_s=120.0, amplitude=50.0)
    cu8 = np.concatenate([sig[: 1 << 17], np.full(1 << 21, 128, np.uint8), sig[1 << 17:]])
    for flags in (["-v"], ["-p", "T", "-v"]):
        want = oracle.run(cu8, flags_to_oracle_opts(oracle, flags))["text"]
        with wm.Receiver(n_streams=1, max_push_bytes=1 << 20, **flags_to_kwargs(flags)) as rx:
            assert rx.run(cu8, push_bytes=1 << 20)[0] == want
        assert len(want.splitlines()) >= 2


def test_partial_tail_and_empty_input(wm, samples):
    cu8 = samples["samples2"]
    with wm.Receiver(n_streams=1, max_push_bytes=4 << 20) as rx:
        a = rx.run(cu8[: 300 * 4096 + 4095])[0]
    with wm.Receiver(n_streams=1, max_push_bytes=4 << 20) as rx:
        assert rx.run(cu8[: 300 * 4096])[0] == a
        assert rx.run(cu8[:100]) == [""]


def test_cli_is_a_drop_in(wm, samples):
    """stdin cu8 -> stdout lines through the plain-C CLI, same switches as the reference."""
    env = dict(os.environ, WMBUS_FIXED_TS="1")
    for name, flags in (BUNDLED_CASES[0], BUNDLED_CASES[1], BUNDLED_CASES[6], BUNDLED_CASES[12]):
        cu8 = np.fromfile(os.path.join(SAMPLES, name), np.uint8)
        p = subprocess.run([wm.CLI_PATH] + flags + ["-B", str(1 << 19)], input=cu8.tobytes(), capture_output=True, env=env)
        assert p.returncode == 0, p.stderr
        assert p.stdout.decode() == BUNDLED[f"{name}|{' '.join(flags)}"]


def test_cli_batch_mode_and_tcp_input(wm, oracle, samples, tmp_path):
    """Extensions of the CLI (SURVEY 8(f) ingest side): several cu8 files decoded in lock step on one
    GPU through double-buffered pinned staging, and cu8 over TCP.  Each file's lines must be exactly
    what the same capture gives alone (= the reference's stdout)."""
    import socket, threading
    env = dict(os.environ, WMBUS_FIXED_TS="1")
    caps = {"a.cu8": samples["samples2"],
            "b.cu8": wm.synth_capture(seed=501, n_samples=3 << 18, kinds=15, frames_per_s=80.0)[0],     # longer than a
            "c.cu8": samples["samples2"][: 150 * 4096 + 17]}                                            # shorter, ragged tail
    want = {}
    for name, cu8 in caps.items():
        cu8.tofile(tmp_path / name)
        want[name] = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]))["text"]
    p = subprocess.run([wm.CLI_PATH, "-v", "-B", str(1 << 19)] + list(caps), cwd=tmp_path, capture_output=True, env=env)
    assert p.returncode == 0, p.stderr
    got = {name: "" for name in caps}
    for line in p.stdout.decode().splitlines(True):
        name, rest = line.split(": ", 1)
        got[name] += rest
    assert got == want
    assert BUNDLED[f"{S2_NAME}|-v"] == want["a.cu8"]

    srv = socket.socket(); srv.bind(("127.0.0.1", 0)); srv.listen(1)
    port = srv.getsockname()[1]

    def serve():
        c, _ = srv.accept()
        c.sendall(samples["samples2"].tobytes()); c.close()
    th = threading.Thread(target=serve); th.start()
    p = subprocess.run([wm.CLI_PATH, "-v", "-T", f"127.0.0.1:{port}"], capture_output=True, env=env, stdin=subprocess.DEVNULL)
    th.join(); srv.close()
    assert p.returncode == 0, p.stderr
    assert p.stdout.decode() == want["a.cu8"]


def test_full_size_batch_matches_the_oracle(wm, oracle):
    """BASELINE configs[3] shape at an eighth of its width: 128 captures (two waves of 64: the clock kernel's cooperative
    path, exactly one of bench.py's eight contexts) x 2^22 IQ samples, one push.  EVERY capture's text against the oracle
    (farmed over the host's cores), plus the properties the synthetic recipe offers: every complete strong burst comes
    out CRC-ok, nothing CRC-ok comes out that was not sent, and the result does not depend on the segmentation."""
    n_streams, n = 128, 1 << 22
    caps, sent = [], []
    for s in range(n_streams):
        c, fr = wm.synth_capture(seed=0xC0FFEE + s, n_samples=n, kinds=7, frames_per_s=20.0)
        caps.append(c); sent.append(fr)
    want = oracle.run_many(caps, oracle.make_opts())
    with wm.Receiver(n_streams=n_streams, max_push_bytes=2 * n) as rx:          # the default tuning, as bench.py runs it
        got = rx.run(caps)
        tim = rx.timing()
    assert got == want
    assert tim["warnings"] == 0
    if os.environ.get("WMBUS_TEST_ROUNDS_ON_HOST", "0") != "1":       # the unattended rounds suffice for the bench workload
        assert tim["slow_path"] == 0
    with wm.Receiver(n_streams=32, max_push_bytes=2 * n, seg_len=16384, rla_seg_len=2048) as rx:
        assert rx.run(caps[:32]) == want[:32]
    n_good = 0
    for s in range(n_streams):
        good = {l.split(";")[-1][2:] for l in got[s].splitlines() if l.split(";")[2] == "1"}
        tx = {f["telegram"].hex() for f in sent[s]}
        assert good <= tx
        assert all(f["telegram"].hex() in good for f in sent[s] if f["complete"])
        n_good += len(good)
    assert n_good > 20 * n_streams


def test_c3_configuration_at_its_survey_size(wm, oracle):
    """BASELINE configs[2] / SURVEY 8 C3: 4.0 MS/s, -d 5 -s, S1 + T1 + C1 chains concurrently, 2^22 IQ samples, all
    four frame kinds placed at +-325 kHz; text, soft symbols, RSSI, slicer bits and every chip against the oracle."""
    flags = ["-d", "5", "-s", "-v"]
    cu8 = wm.synth_capture(seed=20260, n_samples=1 << 22, fs_khz=4000, kinds=15, frames_per_s=60.0, amplitude=60.0,
                           t1c1_center_khz=325.0, s1_center_khz=-325.0)[0]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, flags), taps=True, chips=True)
    modes = {l.split(";")[1] for l in ref["text"].splitlines() if l.split(";")[2] == "1"}
    assert modes == {"T1", "C1", "S1"} and len(ref["text"].splitlines()) > 60
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, **flags_to_kwargs(flags)) as rx:
        assert rx.run(cu8)[0] == ref["text"]
        compare_taps(rx, ref)
        compare_chips(rx, ref)
    with wm.Receiver(n_streams=1, max_push_bytes=1 << 20, **flags_to_kwargs(flags)) as rx:
        assert rx.run(cu8, push_bytes=1 << 20)[0] == ref["text"]           # eight pushes: 104 857.6 decimated samples each, ragged


def test_cli_unique_and_crc_only_options(wm, samples):
    """-U (twin de-duplication) and -W (that plus CRC-clean telegrams only): extensions, off by default."""
    env = dict(os.environ, WMBUS_FIXED_TS="1")
    cu8 = samples["samples2"].tobytes()
    base = BUNDLED[f"{S2_NAME}|-v"].splitlines(True)
    out = {}
    for opt in ("-U", "-W"):
        p = subprocess.run([wm.CLI_PATH, "-v", opt], input=cu8, capture_output=True, env=env)
        assert p.returncode == 0, p.stderr
        out[opt] = p.stdout.decode().splitlines(True)
    assert out["-U"] == [base[0], base[2], base[3]]          # the 71200023 pair is identical: its run-length copy goes
    assert out["-W"] == [base[0]]                             # the 64700082 lines fail their CRC


def test_bench_runs_two_ranks_through_the_hip_library(wm, tmp_path):
    """The N > 1 path of bench.py with the REAL back end (tests/gloo_worker.py, CPU-only, can only stand in the oracle):
    `python bench.py --gpus 2` spawns its two ranks itself; on this one-GPU box both are pointed at device 0 and use
    gloo for the barrier / reductions.  Rank r owns its own captures (seed offset), the JSON line reports n_gpus = 2 and
    the whole-job rate, and the per-rank parity checks (every world-th capture of a rank's first pass: the node's oracle work does not
    grow with the rank count) are reduced over the ranks."""
    import json, sys
    from conftest import ROOT
    env = dict(os.environ, WMBUS_BENCH_BACKEND="gloo", WMBUS_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "128",
                        "--samples", str(1 << 20), "--contexts", "2", "--no-cpu-baseline", "--details", str(tmp_path / "details.json")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])                       # the compact line the driver records ...
    assert len(p.stdout.strip().splitlines()[-1]) < 2000
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["parity"]["ok"] and line["parity"]["first_pass"] == "64/64"
    full = json.load(open(tmp_path / "details.json"))                          # ... and the full record beside it
    assert full["value"] == line["value"]
    assert full["parity"]["ok"] and full["parity"]["ranks"] == 2 and full["parity"]["first_pass"]["captures_compared"] == 64 and full["parity"]["all_ranks"]["first_pass_captures"] == 128
    assert full["datagrams_per_step"] > 100 and len(full["host"]["threads_per_rank"]) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("atan_mode", [1, 2])
@pytest.mark.parametrize("flags,fs,kw", [(["-v"], 1600, {}), (["-v", "-d", "3", "-s"], 2400, dict(t1c1_center_khz=325.0, s1_center_khz=-325.0))])
def test_atan2_approximation_options_match_oracle(wm, oracle, atan_mode, flags, fs, kw):
    """wmbus_cfg.atan_mode (CLI -A): atan2.h's approximations in the discriminator; the oracle's functions are
    pinned to the reference's own (test_oracle_units.py)."""
    cu8 = wm.synth_capture(seed=31 + atan_mode, n_samples=1 << 19, fs_khz=fs, kinds=15, frames_per_s=150.0, amplitude=60.0, **kw)[0]
    oo = flags_to_oracle_opts(oracle, flags)
    oo.atan_mode = atan_mode
    ref = oracle.run(cu8, oo, taps=True, chips=True)
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, atan_mode=atan_mode, **flags_to_kwargs(flags)) as rx:
        assert rx.run(cu8, push_bytes=4096 * 50)[0] == ref["text"]
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, atan_mode=atan_mode, **flags_to_kwargs(flags)) as rx:
        assert rx.run(cu8)[0] == ref["text"]
        compare_taps(rx, ref)
        compare_chips(rx, ref)


def test_abi_details_on_the_device(wm, oracle, samples):
    """wmbus_timing.chips is filled ([chain][algo], the chips of the last push), cfg.keep_taps gates wmbus_read_tap,
    and a full-size push at the largest decimation stays inside the input window (the last partial tile of a push
    stages a whole tile of input: ADVICE r1)."""
    cu8 = samples["samples2"]
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]), chips=True)
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size) as rx:
        rx.run(cu8)
        tim = rx.timing()
        for ch in (0, 1):
            for al in (0, 1):
                assert tim["chips"][ch][al] == int(((ref["chips"]["chain"] == ch) & (ref["chips"]["algo"] == al)).sum())
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, keep_taps=False) as rx:
        rx.run(cu8)
        with pytest.raises(wm.WmbusError):
            rx.read_tap("dphi", 0, 0, 16)
    # d = 16 (run-time decimation kernel), nbytes == max_push_bytes, several streams: the LAST stream's last tile reads
    # the furthest past its staged bytes
    n = 1 << 19
    caps = [wm.synth_capture(seed=7100 + s, n_samples=n, fs_khz=12800, kinds=7, frames_per_s=200.0)[0] for s in range(3)]
    oo = flags_to_oracle_opts(oracle, ["-d", "16", "-v"])
    with wm.Receiver(n_streams=3, max_push_bytes=2 * n, decimation=16) as rx:
        texts = rx.run(caps)
    assert texts == oracle.run_many(caps, oo)


def test_cli_batch_mode_sharded_over_devices(wm, oracle, samples, tmp_path):
    """`-G all` (every device of the box: one here) and `-G 0,0` (two device slots -> two receiver contexts on two
    worker threads, files dealt round robin: the multi-GPU code path on a single GPU): each file's lines are what the
    capture gives alone, whatever slot it landed on."""
    env = dict(os.environ, WMBUS_FIXED_TS="1")
    caps = {f"f{i}.cu8": (samples["samples2"] if i == 0 else
                          wm.synth_capture(seed=620 + i, n_samples=(2 + i % 3) << 17, kinds=15, frames_per_s=90.0)[0]) for i in range(5)}
    want = {}
    for name, cu8 in caps.items():
        cu8.tofile(tmp_path / name)
        want[name] = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]))["text"]
    for g in ("all", "0,0"):
        p = subprocess.run([wm.CLI_PATH, "-v", "-B", str(1 << 19), "-G", g] + list(caps), cwd=tmp_path, capture_output=True, env=env)
        assert p.returncode == 0, p.stderr
        got = {name: "" for name in caps}
        for line in p.stdout.decode().splitlines(True):
            name, rest = line.split(": ", 1)
            got[name] += rest
        assert got == want, g


def _interferer_capture(wm):
    from test_rla_emulated import interferer_capture
    return interferer_capture(wm)


@pytest.mark.parametrize("flags", [["-v"], ["-o", "-v"]], ids=lambda f: " ".join(f))
def test_chip_flood_spills_instead_of_failing(wm, oracle, flags):
    """a square-wave FM interferer makes the reference emit three chips per sample for a while; its chip
    loop never gives up (rtl_wmbus.c:729-803).  The product's run-length segments continue in the spill arena: no push
    fails, chips (read back through the chains) and text equal the oracle's, one-shot and streamed."""
    cu8 = _interferer_capture(wm)
    ref = oracle.run(cu8, flags_to_oracle_opts(oracle, flags), taps=True, chips=True)
    oc = ref["chips"][(ref["chips"]["chain"] == 0) & (ref["chips"]["algo"] == 0)]
    if "-o" not in flags:                                                # (the DC remover tames this interferer: 1 037 chips per segment at most)
        assert np.bincount(oc["sample"] // 8192).max() > 2 * 8192        # far more than a primary region (half a chip per sample) holds
    kw = flags_to_kwargs(flags)
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, **kw) as rx:
        assert rx.run(cu8)[0] == ref["text"]
        assert rx.timing()["warnings"] == 0
        compare_chips(rx, ref)
    with wm.Receiver(n_streams=1, max_push_bytes=1 << 18, **kw) as rx:
        assert rx.run(cu8, push_bytes=1 << 18)[0] == ref["text"]


def test_two_mode_capture_with_d3_s_o_a_spills(wm, oracle):
    """The switch combination the emulation campaign found (-d 3 -s -o -a on a capture carrying both modes): the
    bit-length tracker collapses without any interferer (3.1 chips per sample for a while)."""
    flags = ["-d", "3", "-s", "-o", "-a", "-v"]
    n_chips = 0
    for seed in (3, 4, 5):
        cu8 = wm.synth_capture(seed=880 + seed, n_samples=1 << 19, fs_khz=2400, kinds=15, frames_per_s=150.0, amplitude=60.0,
                               t1c1_center_khz=325.0, s1_center_khz=-325.0)[0]
        ref = oracle.run(cu8, flags_to_oracle_opts(oracle, flags), taps=True, chips=True)
        oc = ref["chips"][(ref["chips"]["chain"] == 0) & (ref["chips"]["algo"] == 0)]
        for seg in (0, 1024):                                  # 1024-sample segments: a primary region of 520 chips, outgrown here
            with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, rla_seg_len=seg, **flags_to_kwargs(flags)) as rx:
                assert rx.run(cu8)[0] == ref["text"]
                assert rx.timing()["warnings"] == 0
                compare_chips(rx, ref)
        n_chips += int(np.bincount(oc["sample"] // 1024).max() > 520)
    assert n_chips > 0                                         # the spill path really ran


def test_exhausted_spill_arena_is_a_warning_not_an_error(wm, oracle):
    """A spill arena of two chunks cannot hold the flood: chips are dropped and wmbus_timing.warnings says so, but the
    push succeeds, the context stays usable and -- every carried state being exact -- the following pushes are the
    oracle's text again."""
    cu8 = _interferer_capture(wm)
    quiet = wm.synth_capture(seed=4711, n_samples=1 << 18, kinds=15, frames_per_s=120.0, amplitude=60.0)[0]
    both = np.concatenate([cu8, quiet])
    ref = oracle.run(both, flags_to_oracle_opts(oracle, ["-v"]))["text"]
    ref_first = oracle.run(cu8, flags_to_oracle_opts(oracle, ["-v"]))["text"]
    with wm.Receiver(n_streams=1, max_push_bytes=cu8.size, spill_words=4096) as rx:
        first = rx.push([cu8])
        assert rx.timing()["warnings"] & 1
        second = rx.push([quiet])
        assert rx.timing()["warnings"] == 0
    assert ref.startswith(ref_first) and second == ref[len(ref_first):]
    assert set(first.splitlines()) <= set(ref_first.splitlines())          # nothing invented; lines inside the flood may be lost
