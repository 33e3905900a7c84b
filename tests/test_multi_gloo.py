"""world_size-2 gloo run of the multi-GPU path (no GPU needed): file-per-rank sharding with disjoint
captures, barrier, max-over-ranks timing and summed datagram count."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_two_ranks_shard_without_overlap(tmp_path, wm, oracle):
    out = str(tmp_path / "gloo.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tests", "gloo_worker.py"), out]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.load(open(out))
    assert r["world"] == 2 and len(r["ranks"]) == 2
    seeds = [s for rk in r["ranks"] for s in rk["seeds"]]
    assert len(set(seeds)) == len(seeds)                          # no capture processed twice
    assert r["ranks"][0]["seeds"][0] == 0xC0FFEE                   # rank 0 = the 1-GPU workload
    assert r["total_lines"] == sum(rk["lines"] for rk in r["ranks"]) > 0
    assert abs(r["slowest"] - max(rk["elapsed"] for rk in r["ranks"])) < 1e-6
    owned = [c for rk in r["ranks"] for c in rk["owned"]]
    assert sorted(owned) == list(range(7))                        # block partition covers every file once
