#!/usr/bin/env python3
"""Can ONE host turn eight GPUs' worth of records into lines?  (the 8-GPU run itself is the driver's.)

R processes ("ranks"), each a wmbus_batch of the bench's shape for one GPU (1024 captures x 2^22 IQ samples in 8 contexts -- or
`--streams`), all on device 0: every rank runs two pushes for real, which leaves the records of a push (WmPkt / WmBurstHdr, chips,
bytes) in every context's page-locked result areas.  Then the GPU is left alone: at a common start time every context of every rank
replays the HOST half of wmbus_collect over its records (wmbus_debug_replay_decode: sort, strip / format, merge) for `--seconds`,
each context on a thread of its own with its decoder pool as in the product.  Printed: lines per second per rank and in all, against
the 8 x 3.7 M lines/s eight GPUs would deliver, and the milliseconds one context-push's decode takes while all ranks decode at once
(the product has ~24 ms per push to hide it in).

    python tools/host_replay.py --ranks 8 --streams 1024 --seconds 3
"""
import argparse, ctypes, importlib, json, os, subprocess, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def rank_main(a):
    wm = importlib.import_module("rtl-wmbus_amd")
    shard = importlib.import_module("rtl-wmbus_amd.shard")
    n, S = a.samples, a.streams
    host_threads = shard.host_threads_per_context(a.ranks, 8)
    caps = [wm.synth_capture(seed=0xC0FFEE + a.rank * S + s, n_samples=n, kinds=wm.T1 | wm.C1A | wm.C1B, frames_per_s=20.0)[0] for s in range(min(S, 64))]
    with wm.Batch(n_streams=S, max_push_bytes=2 * n, host_threads=host_threads) as b:
        for s in range(S):
            b.stage(s, caps[s % len(caps)])
        b.run_resident(2 * n, 2, want_lines=False)
        L = wm.lib()
        ctxs = [rx for rx, _f, _c in b.contexts]
        # wait for the common start (a file all ranks poll: no GPU, no sockets)
        open(os.path.join(a.dir, f"ready{a.rank}"), "w").close()
        while len([f for f in os.listdir(a.dir) if f.startswith("ready")]) < a.ranks:
            time.sleep(0.01)
        res = [None] * len(ctxs)

        def work(i):
            lines, secs, reps = 0, 0.0, 0
            t_end = time.perf_counter() + a.seconds
            while time.perf_counter() < t_end:
                s = ctypes.c_double()
                k = L.wmbus_debug_replay_decode(ctxs[i]._h, 4, ctypes.byref(s))
                if k < 0:
                    raise RuntimeError(f"replay failed: {k}")
                lines += 4 * k; secs += s.value; reps += 4
            res[i] = (lines, secs, reps)
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        wall = time.perf_counter() - t0
    out = {"rank": a.rank, "contexts": len(ctxs), "host_threads_per_context": host_threads, "lines": sum(r[0] for r in res), "wall_s": wall,
           "lines_per_push_per_context": res[0][0] // max(1, res[0][2]), "decode_ms_per_context_push": round(1e3 * sum(r[1] for r in res) / max(1, sum(r[2] for r in res)), 3)}
    print("RANK " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=1 << 22)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--dir", default="")
    a = ap.parse_args()
    if a.rank >= 0:
        return rank_main(a)
    import tempfile
    d = tempfile.mkdtemp(prefix="wm_replay_")
    procs = [subprocess.Popen([sys.executable, __file__, "--ranks", str(a.ranks), "--streams", str(a.streams), "--samples", str(a.samples), "--seconds", str(a.seconds),
                               "--rank", str(r), "--dir", d], stdout=subprocess.PIPE, text=True) for r in range(a.ranks)]
    rows = []
    for p in procs:
        o, _ = p.communicate()
        rows += [json.loads(ln[5:]) for ln in o.splitlines() if ln.startswith("RANK ")]
    if len(rows) != a.ranks:
        print("some ranks failed", file=sys.stderr); sys.exit(1)
    total = sum(r["lines"] / r["wall_s"] for r in rows)
    print(json.dumps({"ranks": a.ranks, "captures_per_rank": a.streams, "cpus": os.cpu_count(), "host_threads_per_context": rows[0]["host_threads_per_context"],
                      "lines_per_s_all_ranks": round(total), "needed_for_8_gpus_lines_per_s": 8 * 3_700_000,
                      "lines_per_context_push": rows[0]["lines_per_push_per_context"],
                      "decode_ms_per_context_push": [r["decode_ms_per_context_push"] for r in sorted(rows, key=lambda r: r["rank"])]}))


if __name__ == "__main__":
    main()
