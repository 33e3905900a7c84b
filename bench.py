#!/usr/bin/env python3
"""bench.py -- cu8 IQ -> datagrams throughput of the MI355X back end (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[3], per GPU): 1024 independent synthetic 1.6 MS/s cu8 captures of
2^22 IQ samples each (SURVEY.md 8(d) recipe: noise + T1/C1 bursts, ~20 bursts/s), resident in HBM
before the timed region.  One "step" = one pass of the whole hot path over that batch: demodulation,
clock recovery, both framers, burst extraction, D2H of the bursts and the host packet decoders
(datagram text produced).  The batch is a wmbus_batch of the LIBRARY (include/wmbus_hip.h; the same
object `rtl_wmbus_hip FILE...` runs): eight receiver contexts of 128 captures that free-run through
their K passes on the library's worker threads.  With N > 1 every rank owns its own 1024 captures on its own GPU
(file-per-GPU sharding, no data-path collective): weak scaling; torch.distributed (RCCL) is used
only for the barrier and the max-over-ranks of the elapsed time.

The JSON line also carries
  roofline     -- for the dominant kernel (k1_demod2): algorithmic bytes (2 B per input IQ sample)
                  per launch / its HIP-event duration measured on the library's own stream (one
                  extra pass per context ALONE after the timed region: inside it a launch shares the
                  GPU with the other contexts' kernels), against 8 TB/s HBM; `traffic` = HBM bytes
                  per launch from the committed rocprofv3 PMC pass (profiles/), null if absent;
  cpu_baseline -- the unmodified reference (oracle/_ref/rtl_wmbus) timed on this host's cores on a
                  bounded sample of the same captures (rank 0, N = 1 only).
"""
import os

# GPU_MAX_HW_QUEUES: libwmbus_hip.so sets its own default (16) when it is loaded -- but in a multi-rank run torch
# initialises the HIP runtime first (shard.init: torch.cuda.set_device for RCCL), and the runtime reads the variable only
# once.  So the same default is put in place here, before anything can have started HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import argparse
import collections
import concurrent.futures as cf
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy rate)
BYTES_PER_SAMPLE = 2            # one u8 I + one u8 Q, read once


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=1024, help="captures per GPU")
    ap.add_argument("--samples", type=int, default=1 << 22, help="IQ samples per capture per step")
    ap.add_argument("--seg-len", type=int, default=0)
    ap.add_argument("--rla-seg-len", type=int, default=0)
    ap.add_argument("--warmup-s1", type=int, default=0)
    ap.add_argument("--warmup-t1c1", type=int, default=0)
    ap.add_argument("--rla-lookback", type=int, default=0)
    ap.add_argument("--contexts", type=int, default=0, help="receiver contexts per GPU (0: the library's default split, 8 x 128 captures)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: ranks only shard, meet at the barriers and reduce (CPU test of the N > 1 path)")
    ap.add_argument("--from-host", action="store_true",
                    help="informational: stage every capture from pinned host memory inside each step (PCIe-inclusive rate; "
                         "the BASELINE metric keeps the input resident in HBM)")
    ap.add_argument("--host-threads", type=int, default=0, help="packet-decoder threads per context (0: cores / (ranks x contexts), at most 32)")
    ap.add_argument("--tolerance-mode", action="store_true", help="informational: wmbus_cfg.tolerance_mode = 1 (soft symbols within 2e-6, not bit-identical); never the headline value")
    ap.add_argument("--no-tolerance-leg", action="store_true", help="skip the informational tolerance-mode leg behind the main measurement")
    ap.add_argument("--seed-offset", type=int, default=0, help="other synthetic captures than the headline batch (capture s gets seed 0xC0FFEE + offset + s); evidence runs only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rla", action="store_true", help="evidence runs only (with --quick): -r 0, the run-length framer off")
    ap.add_argument("--no-time2", action="store_true", help="evidence runs only (with --quick): -t 0, the time2 framer off (clock recovery still runs)")
    ap.add_argument("--no-legs", action="store_true", help="skip the informational legs behind the main measurement (configs[1] / configs[2], the CLI's own rate)")
    ap.add_argument("--quick", action="store_true", help="A/B runs: --no-check --no-cpu-baseline --no-legs --no-tolerance-leg")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B runs: a tuning field of wmbus_cfg for the main batch (k1_small_tile=1, rssi_full=1, rssi_dense_pm=500, rounds_on_host=1 ...); repeatable")
    ap.add_argument("--details", default=os.path.join(ROOT, "gpurun_out", "bench_details.json"),
                    help="where the full record goes (per-context stage times, every leg with its sub-measurements); the stdout line is the compact one")
    return ap.parse_args()


def cpu_baseline(caps, n_samples):
    """The reference binary, one process per core, over whole captures read from /dev/shm;
    bounded to roughly 20 s of wall time."""
    import oracle_ffi as O
    import shutil
    import tempfile
    exe, kind = (O.REF_BIN, "reference") if os.path.exists(O.REF_BIN) else (O.ORACLE_CLI, "port")
    if not os.path.exists(exe):
        return None
    # one reference process per core, at most 64: on the 256-thread bench hosts 256 processes deliver LESS in total (162
    # against 246 Msamples/s, and take 250 s): the aggregate saturates near 64, so that is the baseline's best case
    cores = min(os.cpu_count() or 1, 64)
    d = tempfile.mkdtemp(prefix="wmbus_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        files = []
        for w in range(min(cores, 32)):                       # 32 different captures, shared read-only by the processes
            f = os.path.join(d, f"c{w}.cu8")
            caps[w % len(caps)].tofile(f)
            files.append(f)
        files = [files[w % len(files)] for w in range(cores)]
        t = time.perf_counter()
        subprocess.run(f"{exe} < {files[0]} > /dev/null", shell=True, check=True)
        one = time.perf_counter() - t
        reps = max(1, min(200, int(5.0 / max(one, 1e-3))))      # ~5 s alone, ~20 s with every core busy
        t = time.perf_counter()
        procs = [subprocess.Popen(f"for k in $(seq {reps}); do {exe} < {f} > /dev/null; done", shell=True) for f in files]
        for p in procs:
            p.wait()
        dt = time.perf_counter() - t
    finally:
        shutil.rmtree(d, ignore_errors=True)
    total = cores * reps * n_samples
    return {"value": round(total / dt / 1e6, 2), "unit": "Msamples/s", "cores": cores, "host_cores": os.cpu_count(), "kind": kind,
            "single_core_msamples_s": round(n_samples / one / 1e6, 2),
            "sample": f"{cores} concurrent processes x {reps} passes over a capture of {n_samples} IQ samples "
                      f"({os.path.basename(exe)} -O3, default switches, input from /dev/shm), {dt:.1f} s wall"}


def first_pass_picks(S, rank, world):
    """Captures of a rank's batch whose FIRST pass is compared with the oracle: every world-th one, so that the oracle work
    of a node does not grow with the number of ranks (the ranks of one node share its host cores).  One rank: all of them."""
    return list(range(rank % world, S, world)) if world > 1 else list(range(S))


def last_pass_picks(contexts, world, per_wave=8):
    """Captures whose LAST pass (carried state of every earlier push) is compared: `per_wave` of every 64-capture wave of
    every context at one rank (8 -> 128 of 1024), per_wave / world -- at least one -- with several ranks."""
    k = max(1, -(-per_wave // max(1, world)))
    picks = []
    for first, cnt in contexts:
        for w0 in range(0, cnt, 64):
            w = min(64, cnt - w0)
            picks += sorted({first + w0 + (j * (w - 1)) // max(1, k - 1) for j in range(k)} if w > 1 and k > 1 else {first + w0})
    return picks


def oracle_estimate_s(n_captures, n_samples, passes, threads, per_core_msamples_s=16.0, usable_cores=16):
    """What the parity check will cost on this host (printed so that a log shows where the time went): core-seconds of the
    scalar oracle / the cores the box really gives us (the bench hosts show 256 hardware threads and deliver ~16 cores' worth:
    64 reference processes reach 252 Msamples/s together, one alone 16.4)."""
    return round(n_captures * passes * n_samples / (per_core_msamples_s * 1e6) / max(1, min(threads, usable_cores)), 1)


def leg_single_stream(wm, O, name, n, kw, okw, synth_kw, reps=7):
    """configs[1] / configs[2] (informational): ONE capture of n IQ samples resident in HBM, pushed whole; wall clock per
    push (process + collect: every kernel, the host decode) and the stages' HIP-event times; first push against the oracle."""
    cu8 = wm.synth_capture(n_samples=n, **synth_kw)[0]
    want = O.run(cu8, O.make_opts(**okw))["text"]
    with wm.Receiver(n_streams=1, max_push_bytes=2 * n, keep_taps=False, **kw) as rx:
        got = rx.push([cu8])
        ms, tim = [], None
        for _ in range(reps):
            t = time.perf_counter()
            rx.process(2 * n)
            rx.collect()
            ms.append((time.perf_counter() - t) * 1e3)
            tim = rx.timing()
    med = sorted(ms)[len(ms) // 2]
    return {"workload": name, "samples_per_push": n, "ms_per_push": round(med, 3), "value": round(n / med / 1e3, 1), "unit": "Msamples/s",
            "datagrams": len(want.splitlines()), "parity_ok": got == want,
            "stage_ms": {k: round(tim[k], 3) for k in ("demod_ms", "clock_ms", "rla_ms", "gather_ms", "gpu_total_ms", "host_decode_ms")},
            "reruns": {k: tim[k] for k in ("clock_reruns", "rla_reruns", "ema_retries", "slow_path")}}


def leg_live_latency(wm, O, device, sizes=(1 << 16, 1 << 18, 1 << 20, 1 << 22), reps=9):
    """configs[1] seen from the live path (/root/reference/README.md:64-73, rtl_wmbus.c:1298-1308): ONE 1.6 MS/s
    stream, default switches, pushed in pieces of 2^16 ... 2^22 IQ samples -- wall clock per push (process + collect) and the
    stages' HIP-event times, median over the pushes of a 2^23-sample capture fed `reps` pieces at a time; the whole text of every
    run against the oracle fed the same pieces' worth of input."""
    n_total = 1 << 23
    cu8 = wm.synth_capture(n_samples=n_total, seed=0xC2C2, kinds=wm.T1 | wm.C1A | wm.C1B, frames_per_s=20.0)[0]
    rows = {}
    for n in sizes:
        pushes = min(reps + 1, n_total // n)
        want = O.run(cu8[: 2 * n * pushes], O.make_opts())["text"]
        with wm.Receiver(n_streams=1, max_push_bytes=2 * n, keep_taps=False, device=device) as rx:
            got, ms, tims = "", [], []
            for k in range(pushes):
                t = time.perf_counter()
                rx.stage(0, cu8[2 * n * k: 2 * n * (k + 1)])     # from pageable host memory: what a live reader hands over
                rx.process(2 * n)
                got += rx.collect()
                if k:                                         # the first push also loads code objects and sizes pools
                    ms.append((time.perf_counter() - t) * 1e3)
                    tims.append(rx.timing())
        med = lambda v: round(sorted(v)[len(v) // 2], 3)
        rows[str(n)] = {"ms_per_push": med(ms), "x_real_time": round(n / 1.6e6 / (med(ms) * 1e-3), 1), "parity_ok": got == want,
                        "clock_ms": med([t["clock_ms"] for t in tims]), "rla_ms": med([t["rla_ms"] for t in tims]),
                        "burst_ms": med([t["gather_ms"] for t in tims]), "demod_ms": med([t["demod_ms"] for t in tims]),
                        "host_decode_ms": med([t["host_decode_ms"] for t in tims])}
    return {"workload": "configs[1] live path: one 1.6 MS/s stream, default switches, input staged from host memory per push", "by_samples_per_push": rows}


def leg_c3_batch(wm, O, shard, S, n, device, steps, **tune):
    """configs[2] at batch size (informational): S captures at 4.0 MS/s through `-d 5 -s` (both chains fed from the +-325 kHz
    translation), resident in HBM.  Rate, the demodulation kernel alone, 16 captures' first pass against the oracle."""
    caps = [None] * S
    skw = dict(fs_khz=4000, kinds=15, frames_per_s=50.0, t1c1_center_khz=325.0, s1_center_khz=-325.0)

    def gen(s_):
        caps[s_] = wm.synth_capture(seed=0xC3C3C3 + s_, n_samples=n, **skw)[0]
    with cf.ThreadPoolExecutor(min(os.cpu_count() or 1, 64)) as ex:
        list(ex.map(gen, range(S)))
    b = wm.Batch(n_streams=S, contexts=0, max_push_bytes=2 * n, device=device, decimation=5, simultaneous=True, show_algorithm=True, fixed_timestamp=True,
                 host_threads=shard.host_threads_per_context(1, 8), keep_taps=False, **tune)
    try:
        for s_ in range(S):
            b.stage(s_, caps[s_])
        def texts():
            per = collections.defaultdict(list)
            for rx, first, _c in b.contexts:
                for ln in rx.lines():
                    per[first + ln["stream"]].append(ln["text"])
            return per
        b.run_resident(2 * n, 1)
        per = texts()
        # 128 captures (16, first pass only): sixteen of every context, first pass AND the last one -- this leg is where
        # RSSI on demand pauses itself (31 % of the tiles listed), so the hand-over on demand -> full pass -> on demand of the
        # filter's carried state is inside the pushes that are compared
        picks = last_pass_picks([(first, cnt) for _rx, first, cnt in b.contexts], 1, per_wave=8) if S >= 64 else list(range(S))
        opts = O.make_opts(decimation=5, simultaneous=1)
        want = O.run_many([caps[s_] for s_ in picks], opts)
        bad = [s_ for s_, w in zip(picks, want) if "".join(per[s_]) != w]
        modes = collections.Counter()
        b.run_resident(2 * n, 2)
        tims = []
        t0 = time.perf_counter()
        b.run_resident(2 * n, steps, lambda _f, _c, _l, tm: (tims.append(tm), modes.update([tm["rssi_mode"]])), want_lines=False)
        dt = time.perf_counter() - t0
        passes_c3 = 3 + steps
        per = texts()
        want_l = O.run_many([caps[s_] for s_ in picks], opts, passes=passes_c3)
        bad_l = [s_ for s_, w in zip(picks, want_l) if "".join(per[s_]) != w]
        alone = []
        for rx, _f, _c in b.contexts:
            rx.process(2 * n); rx.collect(); alone.append(rx.timing()["demod_ms"])
        spl = S * n / len(b.contexts)
        k1 = sum(alone) / len(alone)
        mean = {k: round(sum(t[k] for t in tims) / max(1, len(tims)), 3) for k in ("turn_wait_ms", "demod_ms", "clock_ms", "rla_ms", "gather_ms", "gpu_total_ms", "host_decode_ms",
                                                                                  "clock_reruns", "rla_reruns", "ema_retries", "slow_path")}
        for key in ("clock_round", "rla_round"):
            mean[key] = [round(sum(t[key][r] for t in tims) / max(1, len(tims)), 1) for r in range(4)]
        mean["slow_pushes"] = sum(1 for t in tims if t["slow_path"])
        return {"stage_ms_mean_per_context_push": mean,"workload": f"{S} captures x {n} IQ samples at 4.0 MS/s, -d 5 -s, T1 + C1 at +325 kHz and S1 at -325 kHz, HBM-resident",
                "value": round(S * n * steps / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
                "contexts_per_gpu": len(b.contexts), "kernel": "k1_demod2<5, true, false, false>", "k1_alone_ms": round(k1, 3),
                "k1_hbm_frac": round(BYTES_PER_SAMPLE * spl / (k1 / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
                "rssi_modes_timed_pushes": {str(k): v for k, v in sorted(modes.items())},
                "parity": {"captures_compared": len(picks), "mismatches": len(bad), "datagrams": sum(len(w.splitlines()) for w in want),
                           "last_pass": {"pass_number": passes_c3, "captures_compared": len(picks), "mismatches": len(bad_l)}}}
    finally:
        b.close()


def leg_cli(wm, n_files=256, passes=8, distinct=32):
    """The PRODUCT's own command line, measured in THIS run: `rtl_wmbus_hip -v -S f0000.cu8 ...` over n_files capture files of
    passes x 8 MiB in /dev/shm (names are symlinks onto `distinct` different captures), wall-clocked by the program itself:
    decode = wmbus_batch_run (file reads into page-locked slabs, H2D, kernels, host decode, printing); with_setup adds the
    HIP runtime's start, opening the contexts and page-locking the staging."""
    import re
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="wmbus_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        n = passes * (1 << 22)

        def gen(i):
            wm.synth_capture(seed=0xC0FFEE + i, n_samples=n, kinds=wm.T1 | wm.C1A | wm.C1B, frames_per_s=20.0)[0].tofile(os.path.join(d, f"src{i:02d}.cu8"))
        with cf.ThreadPoolExecutor(min(os.cpu_count() or 1, distinct)) as ex:
            list(ex.map(gen, range(distinct)))
        files = []
        for i in range(n_files):
            f = os.path.join(d, f"f{i:04d}.cu8")
            os.symlink(os.path.join(d, f"src{i % distinct:02d}.cu8"), f)
            files.append(f)
        runs = []
        for _rep in range(3):                                     # the first run touches the page cache and the code objects
            t = time.perf_counter()
            p = subprocess.run([wm.CLI_PATH, "-v", "-S"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            wall = time.perf_counter() - t
            m = re.search(rb"total: (\d+) files on (\d+) device\(s\), (\d+) samples; decode ([\d.]+) s = ([\d.]+) Msamples/s; with set-up .*? ([\d.]+) s = ([\d.]+) Msamples/s", p.stderr)
            if p.returncode or not m:
                return {"value": None, "error": f"rc {p.returncode}: {p.stderr[-300:]!r}"}
            runs.append({"decode_s": float(m.group(4)), "decode_msamples_s": float(m.group(5)), "with_setup_s": float(m.group(6)),
                         "with_setup_msamples_s": float(m.group(7)), "process_wall_s": round(wall, 3), "lines": p.stdout.count(b"\n")})
        best = max(runs[1:], key=lambda r: r["decode_msamples_s"])
        return {"command": f"rtl_wmbus_hip -v -S f0000.cu8 ... f{n_files - 1:04d}.cu8 (batch mode, {n_files} files of {passes} x 8 MiB in /dev/shm)", "measured": "in this run",
                "value": best["decode_msamples_s"], "unit": "Msamples/s", "with_setup_msamples_s": best["with_setup_msamples_s"], "samples": n_files * n, "runs": runs,
                "pcie_bound_msamples_s": 24800.0}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def spawn_ranks(n, need_devices=True):
    """`python bench.py --gpus N` without a launcher: become the launcher.  One child per GPU with the
    environment torch.distributed.run would give it (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); rank 0's
    stdout (the JSON line) passes through.  Fails loudly when the node has fewer than N devices."""
    import socket
    if need_devices:
        wm = importlib.import_module("rtl-wmbus_amd")
        have = wm.device_count()
        if have < n and not os.environ.get("WMBUS_BENCH_DEVICE"):
            raise SystemExit(f"bench.py: --gpus {n} but only {have} HIP device(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    import uuid
    tag = uuid.uuid4().hex[:12]                               # this launch's file-group directory (shard.FileGroup)
    for r in range(n):
        env = dict(os.environ, WMBUS_GROUP_TAG=tag, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    raise SystemExit(max(rcs, key=abs))


def main():
    a = parse()
    if a.quick:
        a.no_check = a.no_cpu_baseline = a.no_legs = a.no_tolerance_leg = True
    shard = importlib.import_module("rtl-wmbus_amd.shard")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a.gpus, need_devices=not a.dry_run)
    rank, world, local = shard.rank_env()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # WMBUS_BENCH_BACKEND=gloo|none WMBUS_BENCH_DEVICE=0: several ranks on ONE GPU, to exercise the N > 1
    # code path on a single-GPU box (RCCL refuses two ranks on one device); never set by the driver.  A backend that
    # fails to initialise falls back (nccl -> gloo -> files): the data path needs no collective.
    group = shard.init(world, local, backend=os.environ.get("WMBUS_BENCH_BACKEND") or ("gloo" if a.dry_run else None))   # imports torch BEFORE the HIP library
    S, n = a.streams, a.samples
    if a.dry_run:
        # the N > 1 plumbing without a GPU: every rank names its captures, the ranks meet at the two barriers of the timed
        # region, the reductions run, rank 0 prints ONE line
        seeds = [shard.capture_seed(rank, S, s_) for s_ in (0, S - 1)]
        shard.barrier(group)
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        shard.barrier(group)
        elapsed = shard.max_over_ranks(group, time.perf_counter() - t0)
        allseeds = shard.gather(group, seeds)
        nctx = a.contexts or min(8, max(1, S // 64))
        # the parity sampling of a real run (same functions), with the split the library would make (wmbus_batch_plan needs no device)
        wm_ = importlib.import_module("rtl-wmbus_amd")
        plan, at = [], 0
        for cnt in wm_.batch_plan(S, contexts=a.contexts):
            plan.append((at, cnt)); at += cnt
        fp, lp = first_pass_picks(S, rank, world), last_pass_picks(plan, world)
        counts = shard.gather(group, [len(fp), len(lp)])
        threads = max(1, (os.cpu_count() or 1) // world)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "scaling": "weak", "seed_ranges": allseeds, "elapsed_s": round(elapsed, 4),
                              "backend": getattr(group, "backend", None), "contexts_per_gpu": nctx,
                              "host_threads_per_context": shard.host_threads_per_context(world, nctx),
                              "parity_picks_per_rank": counts, "parity_picks_total": [sum(c[0] for c in counts), sum(c[1] for c in counts)],
                              "generator_threads_per_rank": max(1, min((os.cpu_count() or 1) // world, 64)),
                              "oracle_estimate_s": {"first_pass": oracle_estimate_s(len(fp) * world, n, 1, threads * world),
                                                    "last_pass": oracle_estimate_s(len(lp) * world, n, a.warmup + a.steps + 2, threads * world)}}), flush=True)
        shard.destroy(group)
        return 0
    wm = importlib.import_module("rtl-wmbus_amd")
    if wm.device_count() < 1:
        raise SystemExit("bench.py: no HIP device (the back end has no CPU fallback)")
    local = int(os.environ.get("WMBUS_BENCH_DEVICE", local % wm.device_count()))   # a launcher may narrow the visible devices per rank
    push_bytes = 2 * n

    # ---- synthetic captures (host, multi-threaded), then resident in HBM -------------------------
    t0 = time.perf_counter()
    caps = [None] * S

    def gen(s):
        caps[s] = wm.synth_capture(seed=shard.capture_seed(rank, S, s) + a.seed_offset, n_samples=n, kinds=wm.T1 | wm.C1A | wm.C1B,
                                   frames_per_s=20.0)[0]

    with cf.ThreadPoolExecutor(max(1, min((os.cpu_count() or 1) // world, 64))) as ex:     # the ranks of a node share its cores
        list(ex.map(gen, range(S)))
    t_gen = time.perf_counter() - t0
    t_h2d = time.perf_counter()
    nctx_req = max(0, min(a.contexts, S))
    nctx_guess = nctx_req or min(12 if a.tolerance_mode else 8, max(1, S // 64))
    # host decoder threads per context: ranks x contexts x threads within the host's hardware threads
    host_threads = a.host_threads or shard.host_threads_per_context(world, nctx_guess)
    tune = {}
    for kv in a.tune:
        k_, _, v_ = kv.partition("=")
        tune[k_] = [int(x) for x in v_.split(":")] if ":" in v_ else int(v_)
    batch = wm.Batch(n_streams=S, contexts=nctx_req, max_push_bytes=push_bytes, device=local, seg_len=a.seg_len, rla_seg_len=a.rla_seg_len, **tune,
                     warmup_s1=a.warmup_s1, warmup_t1c1=a.warmup_t1c1, rla_lookback=a.rla_lookback, show_algorithm=True, fixed_timestamp=True,
                     host_threads=host_threads, input_windows=2 if a.from_host else 1, tolerance_mode=int(a.tolerance_mode),
                     rla=not a.no_rla, time2=not a.no_time2)
    nctx = len(batch.contexts)
    per_ctx = [cnt for _, _, cnt in batch.contexts]
    for s in range(S):
        batch.stage(s, caps[s])
    t_h2d = time.perf_counter() - t_h2d      # includes buffer allocation; pageable host memory

    host_caps = None
    if a.from_host:                                    # pinned copies of the captures: the source stages from them itself (zero copy on the host)
        host_caps = wm.pinned_array(S * push_bytes).reshape(S, push_bytes)
        for s_ in range(S):
            host_caps[s_] = caps[s_]

    def run_steps(k_steps, timings=None):
        """K passes of every context over its captures, driven by the library (wmbus_batch_run): the contexts free-run (no
        barrier between steps), a context's next push is on the GPU while its previous one is decoded on the host."""
        on_push = None
        if timings is not None:
            def on_push(first, cnt, _lines, tm):
                timings.setdefault(first, []).append(tm)
        if a.from_host:
            left = {f: k_steps for _, f, _ in batch.contexts}

            def fill(first, cnt, _slab):
                if left[first] == 0:
                    return 0
                left[first] -= 1
                for s_ in range(first, first + cnt):
                    batch.stage(s_, host_caps[s_])            # pinned host memory -> the window the next push reads (copy stream)
                return push_bytes
            return batch.run_from(fill, on_push, self_staged=True, want_lines=False)
        return batch.run_resident(push_bytes, k_steps, on_push, want_lines=False)

    def texts_of_last_push():
        """Datagram text of every capture (index within the rank's batch) as its context's last push printed it."""
        per = collections.defaultdict(list)
        for rx, first, _cnt in batch.contexts:
            for ln in rx.lines():
                per[first + ln["stream"]].append(ln["text"])
        return ["".join(per[s_]) for s_ in range(S)]

    # ---- parity, part 1 (untimed, before the warm-up): the FIRST pass of every context starts from the
    # reference's zero-initialised state, exactly like a fresh oracle run, so EVERY capture of EVERY context is
    # compared with the oracle's text (farmed over this host's cores).
    parity, passes_done = None, 0
    o_threads = max(1, (os.cpu_count() or 1) // max(1, world))
    if not a.no_check:
        import oracle_ffi as O
        run_steps(1)
        passes_done += 1
        got = texts_of_last_push()
        t_o = time.perf_counter()
        fpicks = first_pass_picks(S, rank, world)              # one rank: every capture; N ranks: every N-th of each rank's
        want_by = dict(zip(fpicks, O.run_many([caps[s_] for s_ in fpicks], O.make_opts(), threads=o_threads)))
        want_first = want_by
        bad = [s_ for s_ in fpicks if got[s_] != want_by[s_]]
        parity = {"first_pass": {"captures_compared": len(fpicks), "of": S, "contexts": nctx, "mismatches": len(bad), "first_bad": bad[:4],
                                 "datagrams": sum(len(t.splitlines()) for t in want_by.values()), "oracle_s": round(time.perf_counter() - t_o, 1),
                                 "oracle_estimate_s": oracle_estimate_s(len(fpicks) * world, n, 1, o_threads * world)}}
        if a.tolerance_mode:                                   # informational run: how many LINES differ from the reference's
            dl = 0
            for s_ in bad:
                g, w = collections.Counter(got[s_].splitlines()), collections.Counter(want_by[s_].splitlines())
                dl += sum(((g - w) + (w - g)).values())
            parity["first_pass"]["differing_lines"] = dl
    if a.warmup:
        run_steps(a.warmup)
        passes_done += a.warmup

    shard.barrier(group)
    t0 = time.perf_counter()
    tim_by_ctx = {}
    stats = run_steps(a.steps, tim_by_ctx)
    shard.barrier(group)
    elapsed = time.perf_counter() - t0
    passes_done += a.steps
    try:
        n_threads = len(os.listdir("/proc/self/task"))        # this rank's threads right behind the timed region (workers, decoder pools, runtime)
    except OSError:
        n_threads = 0
    elapsed = shard.max_over_ranks(group, elapsed)
    threads_all = shard.gather(group, [n_threads])
    hd_all = shard.gather(group, [round(sum(tm["host_decode_ms"] for tims in tim_by_ctx.values() for tm in tims) / max(1, sum(len(t_) for t_ in tim_by_ctx.values())), 3)])
    lines_total = int(shard.sum_over_ranks(group, stats["lines"]))
    tim_ctx = [tim_by_ctx.get(first, []) for _, first, _ in batch.contexts]
    demod_ms = sum(tm["demod_ms"] for tims in tim_ctx for tm in tims)
    k1_launches = sum(len(tims) for tims in tim_ctx)
    tim_acc = [[tims[k] for tims in tim_ctx if k < len(tims)] for k in range(a.steps)]

    total_samples = world * S * n * a.steps
    value = total_samples / elapsed / 1e6
    # Roofline of the dominant kernel (k1_demod2).  Inside the timed region the contexts' launches
    # overlap each other and the other kernels, so an event pair around one launch measures its
    # share of the GPU, not its speed.  After the timed region every context therefore makes one more
    # pass ALONE (still HIP events on the library's stream): that duration is the kernel's own.
    samples_per_launch = S * n / nctx
    alone_ms, alone_rssi_ms, alone_mode = [], [], []
    if a.from_host:
        for s_ in range(S):
            batch.stage(s_, host_caps[s_])
    for rx, _first, _cnt in batch.contexts:
        rx.process(push_bytes)
        rx.collect()
        tm_ = rx.timing()
        alone_ms.append(tm_["demod_ms"]); alone_rssi_ms.append(tm_["rssi_ms"]); alone_mode.append(tm_["rssi_mode"])
    passes_done += 1
    k1_avg_s = sum(alone_ms) / max(1, len(alone_ms)) / 1e3
    k1_concurrent_ms = demod_ms / max(1, k1_launches)
    achieved = BYTES_PER_SAMPLE * samples_per_launch / k1_avg_s / 1e9 if k1_avg_s > 0 else 0.0
    on_demand = bool(alone_mode) and all(m_ == wm.RSSI_ON_DEMAND for m_ in alone_mode)
    rssi_avg_ms = sum(alone_rssi_ms) / max(1, len(alone_rssi_ms))
    frac_with_rssi = BYTES_PER_SAMPLE * samples_per_launch / max(k1_avg_s + rssi_avg_ms / 1e3, 1e-12) / 1e9 / HBM_PEAK_GBPS
    traffic, traffic_from = None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            traffic = int(tj["k1_demod_hbm_bytes_per_input_sample"] * samples_per_launch)
            traffic_from = {"file": "profiles/traffic.json", "collected_at": tj.get("collected_at", "unknown"), "job_hbm_bytes_per_step": tj.get("job_hbm_bytes_per_step")}
        except Exception:
            traffic = None

    # ---- parity, part 2: the LAST pass (carried filter / framer / decoder state of every earlier pass) of eight
    # captures of every 64-capture wave of every context (128 of 1024), against an oracle instance that has been
    # fed the same capture the same number of times.
    if parity is not None:
        got = texts_of_last_push()
        picks = last_pass_picks([(first, cnt) for _rx, first, cnt in batch.contexts], world)
        t_o = time.perf_counter()
        want = O.run_many([caps[s_] for s_ in picks], O.make_opts(), passes=passes_done, threads=o_threads)
        bad = [s_ for s_, w in zip(picks, want) if got[s_] != w]
        parity["last_pass"] = {"captures_compared": len(picks), "contexts": nctx, "pass_number": passes_done, "mismatches": len(bad),
                               "first_bad": bad[:4], "oracle_s": round(time.perf_counter() - t_o, 1),
                               "oracle_estimate_s": oracle_estimate_s(len(picks) * world, n, passes_done, o_threads * world)}
        n_bad = int(shard.sum_over_ranks(group, parity["first_pass"]["mismatches"] + len(bad)))
        compared = shard.gather(group, [parity["first_pass"]["captures_compared"], len(picks)])
        parity["ranks"] = world
        parity["all_ranks"] = {"first_pass_captures": sum(c[0] for c in compared), "last_pass_captures": sum(c[1] for c in compared)}
        parity["ok"] = n_bad == 0

    # VALU roofline (BASELINE.md section 3 asks for it next to the HBM one; it is the roof that binds): wave-instructions per
    # input sample from the committed SQ_INSTS_VALU pass (profiles/valu.json) x the samples of a launch / the HIP-event
    # duration measured live above; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction.
    valu = None
    vf = os.path.join(ROOT, "profiles", "valu.json")
    if os.path.exists(vf) and k1_avg_s > 0:
        try:
            vj = json.load(open(vf))
            peak = float(vj["peak_T_wave_instr_per_s"])
            wi = vj["k1_valu_wave_instr_per_input_sample"] * samples_per_launch
            wj = vj["job_valu_wave_instr_per_input_sample"] * S * n
            valu = {"kernel": "k1_demod2", "wave_instr_per_launch": int(wi), "achieved_T_per_s": round(wi / k1_avg_s / 1e12, 4), "peak": peak,
                    "frac": round(wi / k1_avg_s / 1e12 / peak, 4),
                    "whole_job": {"wave_instr_per_step": int(wj), "achieved_T_per_s": round(wj / (elapsed / a.steps) / 1e12, 4),
                                  "frac": round(wj / (elapsed / a.steps) / 1e12 / peak, 4)},
                    "collected_at": vj.get("collected_at", "unknown"),
                    "how": "SQ_INSTS_VALU per launch from the committed rocprofv3 --pmc pass (profiles/valu.json, profiles/*_pmc_sq_counters.csv) / "
                           "the HIP-event duration of this run; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction"}
        except Exception:
            valu = None

    # ---- informational leg (N = 1): the same workload with wmbus_cfg.tolerance_mode = 1 (polynomial arctangent, FMA
    # low-passes; soft symbols within 2e-6 of the reference's instead of bit-identical; BASELINE north_star allows that as
    # long as datagram bytes stay identical).  EVERY capture's first pass is compared with the same oracle texts; the
    # number of differing lines is reported next to the rate.  Never the headline value.
    tol = None
    if world == 1 and parity is not None and not a.tolerance_mode and not a.no_tolerance_leg and not a.from_host:
        try:
            batch.close()
            b2 = wm.Batch(n_streams=S, contexts=0, max_push_bytes=push_bytes, device=local, show_algorithm=True, fixed_timestamp=True,
                          host_threads=shard.host_threads_per_context(1, 12), tolerance_mode=1)
            for s_ in range(S):
                b2.stage(s_, caps[s_])
            b2.run_resident(push_bytes, 1)
            per = collections.defaultdict(list)
            for rx, first, _cnt in b2.contexts:
                for ln in rx.lines():
                    per[first + ln["stream"]].append(ln["text"])
            want1 = want_first                                  # the oracle's text of every capture's first pass, computed above
            dl = dc = 0
            for s_ in sorted(want1):
                g = "".join(per[s_])
                if g != want1[s_]:
                    dc += 1
                    cg, cw = collections.Counter(g.splitlines()), collections.Counter(want1[s_].splitlines())
                    dl += sum(((cg - cw) + (cw - cg)).values())
            b2.run_resident(push_bytes, max(1, a.warmup))
            t0 = time.perf_counter()
            b2.run_resident(push_bytes, a.steps)
            dt = time.perf_counter() - t0
            # the mode's LAST pass (carried filter / framer / decoder state of every push so far) of eight captures of every
            # 64-capture wave (128 of 1024, like the exact mode's check: VERDICT r5 #5; it was two), against an oracle instance fed the same capture the same number of times (the
            # carried-state check of this mode only existed in builder-kept files)
            tol_passes = 1 + max(1, a.warmup) + a.steps
            per = collections.defaultdict(list)
            for rx, first, _cnt in b2.contexts:
                for ln in rx.lines():
                    per[first + ln["stream"]].append(ln["text"])
            tpicks = last_pass_picks([(first, cnt) for _rx, first, cnt in b2.contexts], 1, per_wave=8)
            t_o = time.perf_counter()
            wantl = O.run_many([caps[s_] for s_ in tpicks], O.make_opts(), passes=tol_passes, threads=o_threads)
            ldl = ldc = 0
            for s_, w in zip(tpicks, wantl):
                g = "".join(per[s_])
                if g != w:
                    ldc += 1
                    cg, cw = collections.Counter(g.splitlines()), collections.Counter(w.splitlines())
                    ldl += sum(((cg - cw) + (cw - cg)).values())
            tol_last = {"captures_compared": len(tpicks), "pass_number": tol_passes, "captures_with_differing_text": ldc, "differing_lines": ldl,
                        "datagrams": sum(len(w.splitlines()) for w in wantl), "oracle_s": round(time.perf_counter() - t_o, 1)}
            alone = []
            for rx, _f, _c in b2.contexts:
                rx.process(push_bytes); rx.collect(); alone.append(rx.timing()["demod_ms"])
            spl = S * n / len(b2.contexts)
            tol = {"value": round(S * n * a.steps / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt / a.steps * 1e3, 2), "contexts_per_gpu": len(b2.contexts),
                   "captures_compared": len(want1), "captures_with_differing_text": dc, "differing_lines": dl, "datagrams": sum(len(t.splitlines()) for t in want1.values()),
                   "last_pass": tol_last, "soft_symbol_tolerance_abs": 2e-6,
                   "k1_alone_ms": round(sum(alone) / len(alone), 3),
                   "k1_hbm_frac": round(BYTES_PER_SAMPLE * spl / (sum(alone) / len(alone) / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
                   "what": "wmbus_cfg.tolerance_mode = 1 (CLI -F): polynomial arctangent (|error| <= 2.4e-7 of a half turn) + FMA low-passes in the demodulation "
                           "kernel; RSSI, clock recovery, framers and decoders unchanged.  Opt-in; the headline `value` is the bit-exact default."}
            b2.close()
            batch = None
        except Exception as e:                                # the informational leg must never cost the bench line
            tol = {"value": None, "error": repr(e)}

    # ---- informational legs (N = 1): BASELINE configs[1] and configs[2] -- one 1.6 MS/s stream; one stream and the batch at
    # 4.0 MS/s through `-d 5 -s` -- and the product's own command line.  Each in its own try: never the headline, never fatal.
    legs = {}
    if world == 1 and not a.no_legs and not a.from_host and not a.tolerance_mode:
        import oracle_ffi as O
        if batch is not None:
            batch.close()
            batch = None
        for key, fn in (
            ("c2_single_stream", lambda: leg_single_stream(wm, O, "configs[1]: one 1.6 MS/s capture, default switches (T1/C1 + S1 chains), HBM-resident", n,
                                                           dict(device=local), {}, dict(seed=0xC2C2, kinds=wm.T1 | wm.C1A | wm.C1B, frames_per_s=20.0))),
            ("c3_single_stream", lambda: leg_single_stream(wm, O, "configs[2]: one 4.0 MS/s capture, -d 5 -s, S1 + T1 + C1 concurrently, HBM-resident", n,
                                                           dict(device=local, decimation=5, simultaneous=True), dict(decimation=5, simultaneous=1),
                                                           dict(seed=0xC3C3, fs_khz=4000, kinds=15, frames_per_s=50.0, t1c1_center_khz=325.0, s1_center_khz=-325.0))),
            ("live_latency", lambda: leg_live_latency(wm, O, local)),
            ("c3_batch", lambda: leg_c3_batch(wm, O, shard, S, n, local, max(3, a.steps // 4))),
            ("cli", lambda: leg_cli(wm)),
            ("cli_1024", lambda: leg_cli(wm, n_files=1024)),
        ):
            try:
                t_l = time.perf_counter()
                legs[key] = fn()
                legs[key]["leg_s"] = round(time.perf_counter() - t_l, 1)
            except Exception as e:
                legs[key] = {"value": None, "error": repr(e)}

    ok = True
    if rank == 0:
        def rnd(t):
            return {k: round(v, 3) if isinstance(v, float) else v for k, v in t.items()}
        out = {
            "metric": "Msamples/s cu8 IQ->datagrams, 1024x1.6MS/s streams; %HBM roofline",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{S} synthetic 1.6 MS/s cu8 captures x {n} IQ samples per GPU, T1+C1 bursts, "
                                   f"default switches (T1/C1 + S1 chains, time2 + run-length framers), HBM-resident input",
                       "streams_per_gpu": S, "samples_per_stream": n, "contexts_per_gpu": nctx, "host_threads_per_context": host_threads,
                       "orchestration": "wmbus_batch_run (libwmbus_hip.so): contexts, worker threads, pipelined host decode and the hardware-queue default live in the library",
                       "parallelism": f"file-per-GPU x{world}, no collective", "group_backend": getattr(group, "backend", None)},
            "hbm_roofline_pct_whole_job": round(100.0 * BYTES_PER_SAMPLE * value * 1e6 / world / 1e9 / HBM_PEAK_GBPS, 3),
            "datagrams_per_step": lines_total // max(1, a.steps),
            "roofline": {"bound": "hbm", "kernel": "k1_demod2", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_from": traffic_from,
                         "algorithmic_bytes_per_launch": int(BYTES_PER_SAMPLE * samples_per_launch),
                         "avg_launch_ms": round(k1_avg_s * 1e3, 3),
                         "launches_timed": len(alone_ms),
                         "in_timed_region": {"avg_launch_ms": round(k1_concurrent_ms, 3), "launches": k1_launches,
                                             "achieved": round(BYTES_PER_SAMPLE * samples_per_launch / max(k1_concurrent_ms, 1e-9) / 1e6, 1),
                                             "frac": round(BYTES_PER_SAMPLE * samples_per_launch / max(k1_concurrent_ms, 1e-9) / 1e6 / HBM_PEAK_GBPS, 4)},
                         "valu": valu,
                         "rssi": "on demand: NOT in the timed launch (a second launch computes it for the tiles the bursts touch)" if on_demand
                                 else "in the timed launch (every sample)",
                         "rssi_launch_ms": round(rssi_avg_ms, 3), "frac_with_rssi_launch": round(frac_with_rssi, 4),
                         "dominant": "k1_demod2: two thirds of the job's VALU instructions (profiles/valu.json); the framer kernels are resident longer "
                                     "(latency-bound, they overlap it: profiles/*_bench_kernel_stats.csv)",
                         "dispatches_per_launch": 2,
                         "how": "HIP events around k1_demod2 on the library's stream, one context at a time after the timed "
                                "region (inside it a launch runs beside the other contexts' kernels: avg %.3f ms each).  A push's first pass is TWO "
                                "dispatches of the kernel -- 94 %% of its tiles, then the rest: the next context's turn starts between them -- and a "
                                "launch here spans both: rocprofv3's per-dispatch average (profiles/*_bench_kernel_stats.csv) x 2 is the figure to "
                                "hold against avg_launch_ms" % k1_concurrent_ms},
            "stage_ms_last_step": [rnd(t) for t in (tim_acc[-1] if tim_acc else [])],
            "stage_ms_mid_step": [rnd(t) for t in (tim_acc[a.steps // 2] if tim_acc else [])],
            "setup_s": {"generate": round(t_gen, 1), "alloc_and_h2d": round(t_h2d, 1)},
            "host": {"cpu_count": os.cpu_count(), "threads_per_rank": [t_[0] for t_ in threads_all], "host_decode_ms_mean_per_rank": [h_[0] for h_ in hd_all],
                     "generator_threads_per_rank": max(1, min((os.cpu_count() or 1) // world, 64)), "oracle_threads_per_rank": o_threads},
            "input": "staged from pinned host memory inside every step (PCIe-inclusive, informational)" if a.from_host else "resident in HBM",
            "tolerance_mode": bool(a.tolerance_mode),
        }
        if tol is not None:
            out["tolerance_mode_leg"] = tol
        for key in ("c2_single_stream", "c3_single_stream", "live_latency", "c3_batch", "cli", "cli_1024"):
            if key in legs:
                out[key] = legs[key]
        if "cli" not in legs:                                  # not measured in this run: say where the number comes from
            cli = os.path.join(ROOT, "profiles", "cli_rate.json")     # tools/bench_cli.sh, wall-clocked on an earlier visit
            if os.path.exists(cli):
                try:
                    cj = json.load(open(cli))
                    out["cli_replayed"] = {"replayed_from": "profiles/cli_rate.json", "collected_at": cj.get("collected_at", "unknown commit"), "value": cj.get("value"),
                                           "unit": cj.get("unit"), "note": "NOT measured in this run"}
                except Exception:
                    pass
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(caps, n)
            except Exception as e:                            # the baseline must never cost the bench line
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "reference", "sample": f"failed: {e!r}"}
        if parity is not None:
            fp, lp = parity["first_pass"], parity["last_pass"]
            out["parity"] = parity
            out["parity_check"] = (f"{fp['captures_compared']} of {fp['of']} captures x {nctx} contexts (first pass) and {lp['captures_compared']} captures "
                                   f"across {nctx} contexts (pass {lp['pass_number']}, carried state) identical to the oracle"
                                   if parity["ok"] else "MISMATCH")
            ok = bool(parity["ok"]) or a.tolerance_mode          # tolerance mode is informational: its differences are reported, not fatal
        # ---- the record in full goes to a side file; stdout carries ONE compact line (the 12 KB line of round 4 was
        # cut off in the driver's record, and with it every leg but the headline)
        try:
            os.makedirs(os.path.dirname(a.details), exist_ok=True)
            with open(a.details, "w") as f_:
                json.dump(out, f_)
            details = os.path.relpath(a.details, ROOT)
        except Exception as e:
            details = f"not written: {e!r}"

        def pick(d, *keys):
            return {k: d[k] for k in keys if isinstance(d, dict) and k in d}
        rl, vl = out["roofline"], out["roofline"].get("valu") or {}
        line = pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
        line["config"] = {"workload": f"{S}x{n} IQ samples, 1.6 MS/s cu8, T1+C1 bursts, default switches, HBM-resident", "contexts_per_gpu": nctx,
                          "parallelism": f"file-per-GPU x{world}, no collective"}
        # the line leads with the launch as it runs IN the timed region (beside the other contexts' kernels) and with the whole job's
        # fraction; the launch measured on its own afterwards stands beside them
        ir = rl["in_timed_region"]
        line["roofline"] = dict(pick(rl, "bound", "kernel"), achieved=ir["achieved"], peak=rl["peak"], unit=rl["unit"], frac=ir["frac"], traffic=rl["traffic"],
                                avg_launch_ms=ir["avg_launch_ms"], measured="HIP events around every launch inside the timed region; a launch = 2 dispatches (94 % of a push's tiles, then the rest)",
                                job_frac=round(out["hbm_roofline_pct_whole_job"] / 100.0, 5),
                                alone=dict(pick(rl, "avg_launch_ms", "achieved", "frac", "rssi_launch_ms", "frac_with_rssi_launch"),
                                           rssi="on demand, not in the timed launch" if on_demand else "in the timed launch"))
        if "cpu_baseline" in out and out["cpu_baseline"]:
            cb = out["cpu_baseline"]
            line["cpu_baseline"] = dict(pick(cb, "value", "unit", "cores", "kind", "single_core_msamples_s"), sample=str(cb.get("sample", ""))[:110])
        if parity is not None:
            fp, lp = parity["first_pass"], parity["last_pass"]
            line["parity"] = {"ok": parity["ok"], "first_pass": f"{fp['captures_compared'] - fp['mismatches']}/{fp['captures_compared']}",
                              "last_pass": f"{lp['captures_compared'] - lp['mismatches']}/{lp['captures_compared']} (pass {lp['pass_number']})"}
        sm = {"hbm_pct_job": out["hbm_roofline_pct_whole_job"], "valu_frac_k1": vl.get("frac"), "valu_frac_job": (vl.get("whole_job") or {}).get("frac")}
        if tol is not None:
            tl = tol.get("last_pass") or {}
            sm.update(tol=tol.get("value"), tol_diff_lines=tol.get("differing_lines"), tol_last_diff=tl.get("differing_lines"), tol_k1_ms=tol.get("k1_alone_ms"))
            if tol.get("error"):
                sm["tol_error"] = tol["error"][:80]
        for key, short in (("c2_single_stream", "c2"), ("c3_single_stream", "c3_single")):
            if key in legs:
                sm[short] = legs[key].get("value")
                sm[short + "_ok"] = legs[key].get("parity_ok", legs[key].get("error", "")[:60] if legs[key].get("error") else None)
        if "live_latency" in legs and "by_samples_per_push" in legs["live_latency"]:
            ll = legs["live_latency"]["by_samples_per_push"]
            sm["c2_ms"] = {k: v["ms_per_push"] for k, v in ll.items()}          # ms per push of 2^16 ... 2^22 samples of one live stream
            sm["c2_ms_ok"] = all(v["parity_ok"] for v in ll.values())
        if "c3_batch" in legs:
            cb3 = legs["c3_batch"]
            sm["c3_batch"] = cb3.get("value")
            if "parity" in cb3:
                p3 = cb3["parity"]
                sm["c3_batch_parity"] = f"{p3['captures_compared'] - p3['mismatches']}/{p3['captures_compared']} first, " \
                                        f"{p3['last_pass']['captures_compared'] - p3['last_pass']['mismatches']}/{p3['last_pass']['captures_compared']} pass {p3['last_pass']['pass_number']}"
                sm["c3_k1_ms"] = cb3.get("k1_alone_ms")
            elif cb3.get("error"):
                sm["c3_batch_error"] = cb3["error"][:80]
        if "cli" in legs:
            sm["cli"] = legs["cli"].get("value"); sm["cli_setup"] = legs["cli"].get("with_setup_msamples_s")
            if legs["cli"].get("error"):
                sm["cli_error"] = legs["cli"]["error"][:80]
        if "cli_1024" in legs:
            sm["cli_1024"] = legs["cli_1024"].get("value"); sm["cli_1024_setup"] = legs["cli_1024"].get("with_setup_msamples_s")
        line["summary"] = sm
        line["details"] = details
        print(json.dumps(line, separators=(",", ":")), flush=True)
    elif parity is not None:
        ok = bool(parity["ok"]) or a.tolerance_mode
    if batch is not None:
        batch.close()
    shard.destroy(group)
    return 0 if ok else 3                                     # a rate whose datagrams differ from the reference's is not a result


if __name__ == "__main__":
    sys.exit(main())
