#!/bin/bash
# Register / LDS / occupancy table of every kernel of libwmbus_hip.so, from the compiler's own remarks
# (-Rpass-analysis=kernel-resource-usage) with the Makefile's flags:  tools/kernel_resources.sh [-DNAME=VALUE ...] > profiles/rNN_kernel_resources.txt
# DESIGN.md quotes its register numbers from this file (prose and binary had drifted apart).
here=$(cd "$(dirname "$0")/.." && pwd)/rtl-wmbus_amd
flags=$(make -s -C "$here" print-hipflags)
/opt/rocm/bin/hipcc $flags "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null "$here/csrc/wm_api.hip" 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for ln in sys.stdin:
    m = re.search(r"remark: (?:\[[^\]]*\] )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
def dem(n):
    try: return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0].replace("(anonymous namespace)::", "")
    except Exception: return n
print("%-44s %5s %5s %5s %8s %8s %8s %6s %5s" % ("kernel", "VGPR", "AGPR", "SGPR", "v-spill", "s-spill", "scratch", "LDS", "occ"))
for r in rows:
    print("%-44s %5s %5s %5s %8s %8s %8s %6s %5s" % (dem(r["name"]).replace("void ", "")[:44], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
          r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
print("\nocc = waves per SIMD the register budget allows (VGPR + AGPR share one 512-entry file per lane); LDS is the static part only (K1 adds dynamic LDS: 18.3 KB at d = 2)")
'
