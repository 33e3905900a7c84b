#!/usr/bin/env python3
"""Static instruction budget of the demodulation kernel per stage, read off the gfx950 assembly the compiler emits
("report an instruction budget per stage from the ISA").  The kernel's stages are separated by s_barrier:
  segment 0 = stage 0 (cu8 -> packed int16 staging, table fill)      segment 1 = stage A (boxcars, 8 discriminators, 8 magnitudes)
  segment 2 = magnitude rows to LDS                                   segment 3 = stage B (all four waves' roles: 46-tap FIR quarter +
                                                                                  either one chain's EMA or half the 11-tap FIR) + certification
Segment 3 holds every role's code once; a wave executes its own role only, so the dynamic count per wave is given
separately from the hardware counter pass (profiles/valu.json).
usage: tools/isa_budget.py > profiles/r04_k1_isa_budget.txt      (compiles wm_api.hip with --save-temps into /tmp)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "rtl-wmbus_amd")
flags = subprocess.run(["make", "-s", "-C", HERE, "print-hipflags"], capture_output=True, text=True).stdout.split()
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-w", "--save-temps=obj", "-c", "-o", os.path.join(d, "wm_api.o"), os.path.join(HERE, "csrc", "wm_api.hip")],
                   check=True, cwd=d)
    asm = open(os.path.join(d, "wm_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read()

GROUPS = [("f32 mul / add / sub (VOP2)", r"v_(mul|add|sub|subrev)_f32_e32$"), ("f32 fma / fmac / mad", r"v_(fma|fmac|mad)_f32"),
          ("f32 VOP3 mul / add / sub", r"v_(mul|add|sub)_f32_e64$"), ("rcp / sqrt / rsq", r"v_(rcp|sqrt|rsq)_f32"),
          ("conversions", r"v_cvt_"), ("compares", r"v_cmp"), ("selects (cndmask)", r"v_cndmask"), ("packed int16", r"v_pk_"),
          ("integer / logic / shifts / perm / bfi", r"v_(add|sub|subrev|addc|lshl|lshr|ashr|and|or|xor|perm|bfi|bfe|mul_u32|mul_lo|mad_u|med3|min|max|alignbit|not|lshl_add|lshl_or|or3|add3|add_lshl|mov_b64|mul_hi)"),
          ("v_mov", r"v_mov_b32")]


def budget(name):
    i = asm.index(name + ":")
    j = asm.index(".Lfunc_end", i)                  # the whole function: a kernel may hold several s_endpgm (early returns)
    seg, out = 0, collections.defaultdict(collections.Counter)
    for ln in asm[i:j].splitlines():
        t = ln.strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            seg = min(seg + 1, 3)                     # the compiler lays part of stage B out behind the last barrier: booked together
            continue
        if op.startswith("v_"):
            for g, pat in GROUPS:
                if re.match(pat, op):
                    out[seg][g] += 1
                    break
            else:
                out[seg]["other VALU (" + op + ")"] += 1
            out[seg]["VALU total"] += 1
        elif op.startswith("s_"):
            out[seg]["SALU total"] += 1
        elif op.startswith("ds_"):
            out[seg]["LDS total"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            out[seg]["VMEM total"] += 1
    return out


NAMES = {0: "stage 0", 1: "stage A", 2: "mag rows", 3: "B (all roles)+certify"}
for title, sym in (("k1_demod2<2, false, false, false, 0>  -- the default switches' first pass, bit-exact, RSSI of every sample (contexts with debug views)", "_Z9k1_demod2ILi2ELb0ELb0ELb0ELi0ELi256EEv6K1Args"),
                   ("k1_demod2<2, false, false, false, 1>  -- the same without the RSSI (round 4: RSSI on demand; all of stage B is every wave's)", "_Z9k1_demod2ILi2ELb0ELb0ELb0ELi1ELi256EEv6K1Args"),
                   ("k1_demod2<2, false, false, false, 1, 512>  -- the same on the 2000-sample tile of 512 threads (round 5: the product's first pass at decimation 2)", "_Z9k1_demod2ILi2ELb0ELb0ELb0ELi1ELi512EEv6K1Args"),
                   ("k1_demod2<2, false, false, false, 2>  -- the RSSI of one listed tile (stage B = bracketed warm-ups + 16 samples, waves 0 and 1 only)", "_Z9k1_demod2ILi2ELb0ELb0ELb0ELi2ELi256EEv6K1Args"),
                   ("k1_demod2<2, false, false, true, 0>   -- tolerance mode (polynomial arctangent, FMA low-passes), RSSI of every sample", "_Z9k1_demod2ILi2ELb0ELb0ELb1ELi0ELi256EEv6K1Args"),
                   ("k1_demod2<2, false, false, true, 1>   -- tolerance mode without the RSSI", "_Z9k1_demod2ILi2ELb0ELb0ELb1ELi1ELi256EEv6K1Args")):
    b = budget(sym)
    print(title)
    keys = ["VALU total"] + [g for g, _ in GROUPS] + sorted({k for s in b.values() for k in s if k.startswith("other")}) + ["SALU total", "LDS total", "VMEM total"]
    print("  %-44s" % "static instructions per thread" + "".join("%20s" % NAMES.get(s, s) for s in sorted(b)))
    for k in keys:
        if any(b[s].get(k) for s in b):
            print("  %-44s" % k + "".join("%20d" % b[s].get(k, 0) for s in sorted(b)))
    print()
print("stage A per thread = 4 decimated samples x 2 chains: 8 discriminators (complex product, two correctly rounded divisions, table-driven range\n"
      "reduction, 11-term polynomial, quadrant fix-up: 71 instructions each in the exact kernel) + 8 magnitudes (exact square root: 13 each) + the packed-\n"
      "int16 boxcars.  Stage B is listed with all four wave roles; dynamically a wave runs 4 x 92 (exact) or 4 x 46 (FMA) instructions of the 46-tap FIR\n"
      "plus EITHER one chain's RSSI EMA (48 steps x 3 + 16 byte packs) OR half of the 11-tap FIR (2 x 4 x 22).  Measured per wave (SQ_INSTS_VALU /\n"
      "SQ_WAVES, profiles/valu.json): see DESIGN.md section 6.\n"
      "RS = 1 (no RSSI): three barriers, so the columns are stage 0, stage A and stage B; stage B holds the 11-tap filter twice (two of the four waves\n"
      "run it before the 46-tap one, two after): a wave executes 4 x 92 + 4 x 22 of it, a thread 85 + 634 + ~470 = ~1 190 in all.\n"
      "RS = 2 (the RSSI of a listed tile): the list loop's own barrier shifts the columns by one; its total is what counts (580 VALU static;\n"
      "waves 2 and 3 skip the EMA).")
