#!/usr/bin/env python3
"""Where the demodulation kernel's cycles go, per stage (VERDICT r4 #1a).  Needs a -DWM_K1_STAMPS build of the library:

    tools/build_variant.sh stamps -DWM_K1_STAMPS
    gpurun -- 'WMBUS_HIP_LIB=$PWD/rtl-wmbus_amd/libwmbus_hip_stamps.so python tools/gpu_k1_stamps.py > gpurun_out/k1_stage_cycles.txt'

Every wave of the first pass (RSSI on demand, `k1_demod2<2, false, false, false, 1, NT>`) reads s_memtime at its stage
boundaries and adds the intervals up on the device (wm_k1_demod.h).  Two situations: one context of 128 captures on its own
(the kernel ALONE on the GPU: nothing else of that context runs while its K1 does) and the bench's eight contexts (K1 beside
the other contexts' framer and burst kernels), each for the 976-sample tile of 256 threads and the 2000-sample tile of 512.
The stamps cost the kernel about a tenth of its speed; proportions are what this is for."""
import concurrent.futures as cf
import ctypes
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("rtl-wmbus_amd")

# dynamic VALU instructions per wave and stage, from the ISA (tools/isa_budget.py; stage B: 4 x 92 + 4 x 22 + addressing; 470 before the whole-vector window loads)
INSTR = {256: (30, 554, 446), 512: (34, 554, 446)}
N = 1 << 22


def captures(S):
    caps = [None] * S

    def gen(s):
        caps[s] = wm.synth_capture(seed=0xC0FFEE + s, n_samples=N, kinds=wm.T1 | wm.C1A | wm.C1B, frames_per_s=20.0)[0]
    with cf.ThreadPoolExecutor(min(64, os.cpu_count() or 1)) as ex:
        list(ex.map(gen, range(S)))
    return caps


def stamps(reset):
    out = (ctypes.c_ulonglong * 8)()
    rc = wm.lib().wmbus_debug_k1_stamps(out, int(reset))
    if rc:
        raise SystemExit(f"wmbus_debug_k1_stamps failed: {rc}")
    return [int(v) for v in out]


def run(caps, S, contexts, passes, nt):
    with wm.Batch(n_streams=S, contexts=contexts, max_push_bytes=2 * N, k1_small_tile=nt == 256) as b:
        for s in range(S):
            b.stage(s, caps[s % len(caps)])
        b.run_resident(2 * N, 1, want_lines=False)            # warm-up: code objects, first pass from the zero state
        stamps(True)
        t0 = time.perf_counter()
        demod = []
        b.run_resident(2 * N, passes, on_push=lambda f, n, recs, tim: demod.append(tim["demod_ms"]), want_lines=False)
        dt = time.perf_counter() - t0
    acc = stamps(True)
    return acc, dt, sum(demod) / max(1, len(demod))


def table(title, acc, dt, demod_ms, S, passes, nt):
    waves, tile0 = acc[5], acc[6]
    blocks = waves / (nt // 64)
    names = ("stage 0: input loads issued -> converted, staged", "wait at barrier 1", "stage A: boxcars, 8 discriminators", "wait at barrier 2",
             "stage B: 46-tap + 11-tap low-pass, stores")
    ins = INSTR[nt]
    per_stage_instr = (ins[0], 0, ins[1], 0, ins[2])
    tot = sum(acc[:5])
    print(f"\n{title}: {S} captures x 2^22 samples, {passes} passes, tile of {4 * nt - 48} samples on {nt} threads; {waves} waves, {blocks:.0f} blocks")
    print(f"  wall {dt * 1e3:.1f} ms ({S * N * passes / dt / 1e9:.1f} Gsamples/s with the stamps in), K1 launch(es) per push {demod_ms:.3f} ms (HIP events)")
    print(f"  {'interval':<52}{'cycles/wave':>12}{'share':>8}{'VALU/wave':>11}{'cycles per VALU':>17}")
    for k in range(5):
        cw = acc[k] / waves
        cpi = f"{cw / per_stage_instr[k]:.2f}" if per_stage_instr[k] else ""
        print(f"  {names[k]:<52}{cw:>12.0f}{100.0 * acc[k] / tot:>7.1f}%{per_stage_instr[k] or '':>11}{cpi:>17}")
    print(f"  {'whole tile (a wave, start to end)':<52}{tot / waves:>12.0f}{'':>8}{sum(per_stage_instr):>11}{tot / waves / sum(per_stage_instr):>17.2f}")
    print(f"  the tile as its first wave saw it: {tile0 / blocks:.0f} cycles; with eight waves per SIMD a wave that owns every eighth issue slot")
    print(f"  would show 8 x the SIMD's cycles per instruction: {tot / waves / sum(per_stage_instr) / 8:.2f} SIMD cycles per VALU instruction if nothing else ran")


def main():
    if not hasattr(wm.lib(), "wmbus_debug_k1_stamps"):
        raise SystemExit("this library was built without -DWM_K1_STAMPS (tools/build_variant.sh stamps -DWM_K1_STAMPS; WMBUS_HIP_LIB=...)")
    wm.lib().wmbus_debug_k1_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    big = "--ring" in sys.argv
    caps = captures(1024 if big else 128)
    print("# s_memtime per stage of the demodulation kernel's first pass (shader cycles; the clock under this load is 2.3 GHz, DESIGN section 10)")
    for nt in (256, 512):
        acc, dt, dm = run(caps, 128, 1, 4, nt)
        table("K1 ALONE (one context)", acc, dt, dm, 128, 4, nt)
    if big:
        for nt in (256, 512):
            acc, dt, dm = run(caps, 1024, 8, 6, nt)
            table("K1 IN THE RING (eight contexts)", acc, dt, dm, 1024, 6, nt)


if __name__ == "__main__":
    main()
