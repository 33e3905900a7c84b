"""Worker for tests/test_multi_gloo.py: the N>1 path of bench.py (sharding + barrier + reductions)
on CPU with the gloo backend; the captures of each rank are decoded by the oracle."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_path = sys.argv[1]
    shard = importlib.import_module("rtl-wmbus_amd.shard")
    wm = importlib.import_module("rtl-wmbus_amd")
    import oracle_ffi as O
    rank, world, local = shard.rank_env()
    dist = shard.init(world, local, backend="gloo")
    S, n = 3, 1 << 17
    seeds = [shard.capture_seed(rank, S, s) for s in range(S)]
    shard.barrier(dist)
    t0 = time.perf_counter()
    lines = 0
    for sd in seeds:
        cu8, _ = wm.synth_capture(seed=sd, n_samples=n, kinds=7, frames_per_s=100.0)
        lines += len(O.run(cu8, O.make_opts())["text"].splitlines())
    time.sleep(0.05 * rank)                        # make the ranks finish at different times
    shard.barrier(dist)
    mine = time.perf_counter() - t0
    slowest = shard.max_over_ranks(dist, mine)
    total = shard.sum_over_ranks(dist, lines)
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(rank=rank, seeds=seeds, lines=lines, elapsed=mine,
                                          owned=list(shard.owned_captures(rank, world, 7))))
    if rank == 0:
        json.dump(dict(ranks=gathered, slowest=slowest, total_lines=total, world=world), open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
