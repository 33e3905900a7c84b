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


def test_cli_file_to_device_map_is_the_documented_one(wm, tmp_path):
    """`rtl_wmbus_hip -G <list> -M FILE...` prints the shard map the batch mode uses (the CLI's own shard_slot(), no
    GPU needed for an explicit list); it must be SURVEY 8(e)'s `s mod n` = shard.device_of."""
    import importlib
    shard = importlib.import_module("rtl-wmbus_amd.shard")
    names = [f"cap{i:02d}.cu8" for i in range(19)]
    for devs in ([0, 1, 2, 3, 4, 5, 6, 7], [2, 5], [3]):
        p = subprocess.run([wm.CLI_PATH, "-G", ",".join(map(str, devs)), "-M"] + names, capture_output=True, text=True, cwd=tmp_path,
                           stdin=subprocess.DEVNULL)
        assert p.returncode == 0, p.stderr
        got = [ln.split(" -> device ") for ln in p.stdout.splitlines()]
        assert [g[0] for g in got] == names
        assert [int(g[1]) for g in got] == [devs[shard.device_of(i, len(devs))] for i in range(len(names))]
    # every device of an 8-GPU node gets a share that differs by at most one file
    per = [sum(1 for i in range(19) if shard.device_of(i, 8) == k) for k in range(8)]
    assert max(per) - min(per) <= 1 and sum(per) == 19
    p = subprocess.run([wm.CLI_PATH, "-G", "1,x", "-M", "a.cu8"], capture_output=True, text=True, stdin=subprocess.DEVNULL)
    assert p.returncode == 1
