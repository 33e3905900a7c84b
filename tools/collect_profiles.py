#!/usr/bin/env python3
"""Copy what tools/gpu_profile.sh left under gpurun_out/prof into profiles/ (tracked) and rebuild
profiles/traffic.json.  usage: tools/collect_profiles.py [round-tag, default r01]"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

COPIES = [("trace/bench_kernel_stats.csv", "bench_kernel_stats.csv"),
          ("trace/bench_domain_stats.csv", "bench_domain_stats.csv"),
          ("bench_under_rocprof.json", "bench_under_rocprof.json"),
          ("trace1/single_kernel_stats.csv", "single_context_kernel_stats.csv"),
          ("single_context_under_rocprof.json", "single_context_under_rocprof.json"),
          ("pmc_fetch_size.csv", "pmc_fetch_size.csv"),
          ("pmc_write_size.csv", "pmc_write_size.csv")]
for src, dst in COPIES:
    p = os.path.join(SRC, src)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, f"{tag}_{dst}"))
        print("copied", src)
    else:
        print("MISSING", src)


# slim copy of the kernel trace of the default bench command (what tools/k1_overlap.py analyses)
tr = os.path.join(SRC, "trace", "bench_kernel_trace.csv")
if os.path.exists(tr):
    cols = ["Queue_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "VGPR_Count", "Workgroup_Size_X", "Grid_Size_X"]
    with open(tr) as f, open(os.path.join(DST, f"{tag}_bench_kernel_trace.csv"), "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=cols)
        w.writeheader()
        for row in csv.DictReader(f):
            row["Kernel_Name"] = row["Kernel_Name"].split("(")[0]
            w.writerow({k: row[k] for k in cols})
    print("copied slim kernel trace")


def per_kernel(name):
    out = {}
    p = os.path.join(SRC, name)
    if not os.path.exists(p):
        return out
    for r in csv.DictReader(open(p)):
        out[r["kernel"]] = (int(r["launches"]), float(r["sum_KB"]))
    return out


fetch, write = per_kernel("pmc_fetch_size.csv"), per_kernel("pmc_write_size.csv")


def pick(d, prefix):
    n = s = 0
    for k, (ln, kb) in d.items():
        base = k.replace("void ", "").split("(")[0].split("<")[0]
        if base == prefix:
            n += ln
            s += kb
    return n, s


n_k1, f_k1 = pick(fetch, "k1_demod2")
_, w_k1 = pick(write, "k1_demod2")
if n_k1:
    S, N = 1024, 1 << 22
    fetch_b = 2.0 * f_k1 * 1024 / n_k1          # FETCH_SIZE reports half of wide coalesced reads on gfx950
    write_b = w_k1 * 1024 / n_k1
    traffic = {
        "kernel": "k1_demod2<2,false>",
        "launch": f"{S} captures x 2^22 IQ samples in one launch (bench.py --contexts 1 --steps 1 --warmup 0: the timed step "
                  f"plus the un-overlapped calibration pass = {n_k1} launches)",
        "FETCH_SIZE_KB_per_launch": round(f_k1 / n_k1, 2),
        "WRITE_SIZE_KB_per_launch": round(w_k1 / n_k1, 2),
        "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM); "
                      "WRITE_SIZE taken as reported (k_fill calibration in the first session: 1.10x of the bytes written)",
        "k1_demod_hbm_bytes_per_launch": fetch_b + write_b,
        "k1_demod_hbm_bytes_per_input_sample": (fetch_b + write_b) / (S * N),
        "algorithmic_bytes_per_input_sample": 2,
        "other_kernels_KB_per_step_as_reported": {},
    }
    for pre in ("k2_clock", "k2_clock_rla", "k2_rla", "k3_scan", "k3_bursts"):
        traffic["other_kernels_KB_per_step_as_reported"][pre + "_fetch"] = round(pick(fetch, pre)[1] / 2, 1)   # two passes per run
        traffic["other_kernels_KB_per_step_as_reported"][pre + "_write"] = round(pick(write, pre)[1] / 2, 1)
    json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
