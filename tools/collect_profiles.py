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
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
# which source the numbers belong to: the commit whose build was measured (argument 2), else HEAD of this checkout
import subprocess
try:
    COLLECTED_AT = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
except Exception:
    COLLECTED_AT = "unknown"

COPIES = [("trace/bench_kernel_stats.csv", "bench_kernel_stats.csv"),
          ("trace/bench_domain_stats.csv", "bench_domain_stats.csv"),
          ("bench_under_rocprof.json", "bench_under_rocprof.json"),
          ("trace1/single_kernel_stats.csv", "single_context_kernel_stats.csv"),
          ("single_context_under_rocprof.json", "single_context_under_rocprof.json"),
          ("trace_c3/c3_kernel_stats.csv", "c3_kernel_stats.csv"),
          ("c3_under_rocprof.json", "c3_under_rocprof.json"),
          ("pmc_fetch_size.csv", "pmc_fetch_size.csv"),
          ("pmc_write_size.csv", "pmc_write_size.csv"),
          ("pmc_sq.csv", "pmc_sq_counters.csv")]
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


def family(kernel):
    """Kernel name -> the family its numbers are booked under, or None to skip: the list kernels (re-run lists) ride with
    their first-pass kernels, the RSSI-on-demand launch of K1 (k1_demod2<.., RS = 2>) is booked as "k1_rssi", the option / repair variant of K1 (k1_demod2<D, SHIFT, GEN = true, ..>: empty list launches in
    the bench) is left out."""
    k = kernel.replace("void ", "").split("(")[0].strip()
    base = k.split("<")[0]
    if base == "k1_demod2":
        args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []
        if len(args) >= 3 and args[2] == "true":
            return None
        if len(args) >= 5 and args[4] == "2":
            return "k1_rssi"                             # RSSI on demand: the listed tiles' launch (the first pass is the RS = 1 instantiation)
    return {"k2_clock_list": "k2_clock", "k2_clock_sys": "k2_clock", "k2_clock_sys_list": "k2_clock", "k2_rla_list": "k2_rla", "k2_finish": "k2_verify"}.get(base, base)    # (round 6: the systolic clock kernels and the merged last verification ride with their families)


def pick(d, prefix):
    n = s = 0
    for k, (ln, kb) in d.items():
        base = family(k)
        if base == prefix:
            n += ln
            s += kb
    return n, s


# ---- VALU roofline inputs: wave-instructions per launch from the SQ counter pass (single context: 1024 captures per launch)
sqp = os.path.join(SRC, "pmc_sq.csv")
if os.path.exists(sqp):
    S, N = 1024, 1 << 22
    per = {}
    for r in csv.DictReader(open(sqp)):
        base = family(r["kernel"])
        if base is None:
            continue                                   # repair / option variant: empty list launches
        d = per.setdefault(base, {"launches": 0, "avg_ms_under_pmc": 0.0})
        d[r["counter"]] = d.get(r["counter"], 0.0) + float(r["sum"])
        if r["counter"] == "SQ_INSTS_VALU":
            d["launches"] += int(r["launches"]); d["avg_ms_under_pmc"] = max(d["avg_ms_under_pmc"], float(r["avg_ms_under_pmc"]))
    steps = 2                                  # bench.py --steps 1 --warmup 0: the timed step + the un-overlapped calibration pass
    valu = {"collected_at": COLLECTED_AT, "profiles_tag": tag,
            "run": f"bench.py --contexts 1 --steps 1 --warmup 0: {S} captures x 2^22 IQ samples per launch, {steps} passes",
            "peak_T_wave_instr_per_s": 1.2288,
            "peak_how": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md: SIMD-32, v_fma_f32 2 cyc)",
            "kernels": {}}
    for k, d in per.items():
        if not any(k.startswith(x) for x in ("k1_demod2", "k1_rssi", "k2_clock", "k2_rla", "k3_", "k2_verify")):
            continue
        valu["kernels"][k] = {"valu_wave_instr_per_step": d.get("SQ_INSTS_VALU", 0) / steps, "salu_per_step": d.get("SQ_INSTS_SALU", 0) / steps,
                              "lds_instr_per_step": d.get("SQ_INSTS_LDS", 0) / steps, "waves_per_step": d.get("SQ_WAVES", 0) / steps,
                              "wave_quad_cycles_per_step": d.get("SQ_WAVE_CYCLES", 0) / steps,
                              "active_valu_quad_cycles_per_step": d.get("SQ_ACTIVE_INST_VALU", 0) / steps,
                              "wait_inst_any_quad_cycles_per_step": d.get("SQ_WAIT_INST_ANY", 0) / steps,
                              "wait_any_quad_cycles_per_step": d.get("SQ_WAIT_ANY", 0) / steps}
    k1 = valu["kernels"].get("k1_demod2", {})
    valu["k1_valu_wave_instr_per_input_sample"] = k1.get("valu_wave_instr_per_step", 0) / (S * N)
    valu["job_valu_wave_instr_per_input_sample"] = sum(v["valu_wave_instr_per_step"] for v in valu["kernels"].values()) / (S * N)
    json.dump(valu, open(os.path.join(DST, "valu.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in valu.items() if k != "kernels"}, indent=1))

n_k1, f_k1 = pick(fetch, "k1_demod2")
_, w_k1 = pick(write, "k1_demod2")
if n_k1:
    S, N = 1024, 1 << 22
    PASSES = 2                                  # bench.py --steps 1 --warmup 0: the timed step + the un-overlapped calibration pass
    fetch_b = 2.0 * f_k1 * 1024 / PASSES        # FETCH_SIZE reports half of wide coalesced reads on gfx950; per PASS over the
    write_b = w_k1 * 1024 / PASSES              # 1024 captures (since round 4 a pass is two launches: the main part and the tail)
    traffic = {
        "collected_at": COLLECTED_AT, "profiles_tag": tag,
        "kernel": "k1_demod2<2,false>",
        "launch": f"{S} captures x 2^22 IQ samples per pass (bench.py --contexts 1 --steps 1 --warmup 0: the timed step "
                  f"plus the un-overlapped calibration pass = {PASSES} passes, {n_k1} launches: main part + tail each)",
        "FETCH_SIZE_KB_per_launch": round(f_k1 / PASSES, 2),
        "WRITE_SIZE_KB_per_launch": round(w_k1 / PASSES, 2),
        "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM); "
                      "WRITE_SIZE taken as reported (k_fill calibration in the first session: 1.10x of the bytes written)",
        "k1_demod_hbm_bytes_per_launch": fetch_b + write_b,
        "k1_demod_hbm_bytes_per_input_sample": (fetch_b + write_b) / (S * N),
        "algorithmic_bytes_per_input_sample": 2,
        "other_kernels_KB_per_step_as_reported": {},
    }
    for pre in ("k1_rssi", "k2_clock", "k2_rla", "k3_scan", "k3_spans", "k3_bursts"):
        traffic["other_kernels_KB_per_step_as_reported"][pre + "_fetch"] = round(pick(fetch, pre)[1] / 2, 1)   # two passes per run
        traffic["other_kernels_KB_per_step_as_reported"][pre + "_write"] = round(pick(write, pre)[1] / 2, 1)
    o = traffic["other_kernels_KB_per_step_as_reported"]
    traffic["job_hbm_bytes_per_step"] = fetch_b + write_b + 1024.0 * sum((2.0 if k.endswith("_fetch") else 1.0) * v for k, v in o.items())
    json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
