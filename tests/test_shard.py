"""The N > 1 plumbing without a GPU: `bench.py --gpus N --dry-run` (ranks shard, meet at the barriers, reduce, one JSON
line) over gloo and over the file group that needs no torch.distributed at all, and the host-thread cap."""
import importlib
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("backend,n", [("gloo", 8), ("none", 8), ("none", 3)])
def test_bench_dry_run_shards_and_reduces(backend, n):
    env = dict(os.environ, WMBUS_BENCH_BACKEND=backend, OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # rank 0 prints ONE line
    r = json.loads(lines[0])
    assert r["n_gpus"] == n and r["scaling"] == "weak" and r["backend"] == backend
    ranges = r["seed_ranges"]
    assert len(ranges) == n and ranges[0][0] == 0xC0FFEE     # rank 0 = the 1-GPU workload
    for a, b in zip(ranges, ranges[1:]):
        assert a[1] + 1 == b[0]                               # disjoint, contiguous: no capture twice, none skipped
    assert r["elapsed_s"] >= 0.01 * n                         # max over ranks of what each rank measured
    # the parity sampling of a real run (bench.first_pass_picks / last_pass_picks): the node's oracle work does not grow with
    # the rank count -- 1024 first-pass captures and >= 128 last-pass captures over ALL ranks, every context of every rank sampled
    per_rank = r["parity_picks_per_rank"]
    assert len(per_rank) == n and r["parity_picks_total"][0] in range(1024, 1024 + n) and all(c[0] in (1024 // n, 1024 // n + 1) for c in per_rank)
    assert all(c[1] >= 16 for c in per_rank) and 128 <= r["parity_picks_total"][1] <= 16 * n + 128
    assert r["generator_threads_per_rank"] * n <= max(os.cpu_count() or 1, n)
    assert r["oracle_estimate_s"]["first_pass"] > 0 and r["oracle_estimate_s"]["last_pass"] > 0
    cpus = os.cpu_count() or 16
    assert n * r["contexts_per_gpu"] * r["host_threads_per_context"] <= max(cpus, n * r["contexts_per_gpu"])


def test_parity_picks_cover_every_context():
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    plan = [(128 * i, 128) for i in range(8)]
    one = bench.last_pass_picks(plan, 1)
    assert len(one) == 128 and len(set(one)) == 128 and all(any(f <= s < f + c for s in one) for f, c in plan)
    eight = bench.last_pass_picks(plan, 8)
    assert len(eight) == 16 and all(sum(f <= s < f + c for s in eight) == 2 for f, c in plan)      # one per 64-capture wave
    assert bench.last_pass_picks([(0, 1)], 1) == [0] and bench.last_pass_picks([(0, 70)], 1, per_wave=2) == [0, 63, 64, 69]
    assert bench.first_pass_picks(1024, 0, 1) == list(range(1024))
    got = sorted(s for r in range(8) for s in bench.first_pass_picks(1024, r, 8))
    assert got == list(range(1024))                           # over the ranks every residue is taken once: 128 per rank


def test_host_threads_stay_within_the_box():
    shard = importlib.import_module("rtl-wmbus_amd.shard")
    assert shard.host_threads_per_context(1, 8, cpus=256) == 16
    assert shard.host_threads_per_context(8, 8, cpus=256) == 4   # 8 ranks x 8 contexts x 4 = 256
    assert shard.host_threads_per_context(8, 8, cpus=32) == 1


def test_torchrun_launch_of_the_dry_run():
    """What the driver does for N > 1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`."""
    env = dict(os.environ, WMBUS_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == 2 and len(r["seed_ranges"]) == 2
