"""The oracle (oracle/wmbus_oracle.c) against the reference: committed golden stdout of the
unmodified reference binary, and -- when oracle/_ref exists -- the binary itself."""
import json
import os

import numpy as np
import pytest

from cases import BUNDLED_CASES, SYNTH_CASES, flags_to_oracle_opts, synth_case_capture
from conftest import GOLDEN, SAMPLES

BUNDLED = json.load(open(os.path.join(GOLDEN, "bundled.json")))
SYNTH = json.load(open(os.path.join(GOLDEN, "synthetic.json")))


@pytest.mark.parametrize("name,flags", BUNDLED_CASES, ids=[f"{n[14:22]}:{' '.join(f)}" for n, f in BUNDLED_CASES])
def test_oracle_matches_reference_on_bundled_captures(oracle, name, flags):
    cu8 = np.fromfile(os.path.join(SAMPLES, name), np.uint8)
    got = oracle.run(cu8, flags_to_oracle_opts(oracle, flags))["text"]
    assert got == BUNDLED[f"{name}|{' '.join(flags)}"]


def test_readme_kpi_two_good_datagrams(oracle, samples):
    # README.md:70-71 of the reference: `grep "[T,C,S]1;1;1" | wc -l` on samples2 -> 2
    text = oracle.run(samples["samples2"], flags_to_oracle_opts(oracle, []))["text"]
    assert sum(l.split(";")[0:3] == ["T1", "1", "1"] for l in text.splitlines()) == 2


@pytest.mark.parametrize("case", SYNTH_CASES, ids=[c["id"] for c in SYNTH_CASES])
def test_oracle_matches_reference_on_synthetic_captures(oracle, wm, case):
    cu8, frames = synth_case_capture(wm, case)
    got = oracle.run(cu8, flags_to_oracle_opts(oracle, case["flags"]))["text"]
    assert got == SYNTH[case["id"]]


def test_synthetic_frames_are_decoded(oracle, wm):
    """Known by construction: every complete strong burst is printed with CRC ok by at least one
    of the two framers, and nothing with CRC ok is printed that was not transmitted."""
    case = SYNTH_CASES[1]
    cu8, frames = synth_case_capture(wm, case)
    text = oracle.run(cu8, flags_to_oracle_opts(oracle, case["flags"]))["text"]
    good = {}
    for l in text.splitlines():
        p = l.split(";")
        if p[2] == "1":
            good.setdefault(p[-1][2:], set()).add(p[0])
    sent = {f["telegram"].hex() for f in frames}
    for f in frames:
        if f["complete"]:
            assert good.get(f["telegram"].hex()), f
    assert set(good) <= sent


def test_partial_tail_block_is_dropped(oracle, samples):
    # rtl_wmbus.c:1301-1308: fread of whole 4096-byte blocks, the tail is ignored
    cu8 = samples["samples2"]
    a = oracle.run(cu8[: 300 * 4096], flags_to_oracle_opts(oracle, ["-v"]))["text"]
    b = oracle.run(cu8[: 300 * 4096 + 4095], flags_to_oracle_opts(oracle, ["-v"]))["text"]
    assert a == b


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "rtl_wmbus")),
                    reason="reference binary not built (no /root/reference on this box)")
def test_oracle_matches_live_reference_binary(oracle, wm):
    for seed in range(6):
        amp = [60, 20, 10, 8, 7, 12][seed]
        cu8, _ = wm.synth_capture(seed=500 + seed, n_samples=1 << 20, kinds=15, frames_per_s=60.0, amplitude=amp)
        for flags in (["-v"], ["-v", "-o"], ["-v", "-a"]):
            assert oracle.run(cu8, flags_to_oracle_opts(oracle, flags))["text"] == oracle.run_reference(cu8, flags)
