"""Regenerates tests/golden/*.json from the UNMODIFIED reference binary (oracle/_ref/rtl_wmbus).

Run in the build container (needs /root/reference to have been compiled by `make -C oracle ref`):
    python tests/golden/make_golden.py
The JSON files map "<capture>|<switches>" -> stdout of the reference with the wall-clock timestamp
field replaced by TS.  Synthetic captures are regenerated from (seed, config) by the test itself.
"""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_ffi as O

from cases import BUNDLED_CASES, SYNTH_CASES, synth_case_capture


def main():
    wm = importlib.import_module("rtl-wmbus_amd")
    out = {}
    for name, flags in BUNDLED_CASES:
        cu8 = np.fromfile(os.path.join(HERE, "samples", name), np.uint8)
        out[f"{name}|{' '.join(flags)}"] = O.run_reference(cu8, flags)
    json.dump(out, open(os.path.join(HERE, "bundled.json"), "w"), indent=1)
    out = {}
    for case in SYNTH_CASES:
        cu8, _ = synth_case_capture(wm, case)
        out[case["id"]] = O.run_reference(cu8, case["flags"])
    json.dump(out, open(os.path.join(HERE, "synthetic.json"), "w"), indent=1)
    print("golden files written")


if __name__ == "__main__":
    main()
