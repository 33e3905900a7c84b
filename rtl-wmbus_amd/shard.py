"""File-per-GPU sharding helpers: captures are independent, so rank r simply owns
captures [r*S, (r+1)*S) and there is no data-path collective.  A process group is used only for the
barrier and for max / sum over ranks of a few scalars: torch.distributed with backend "nccl" (= RCCL)
on GPUs or "gloo" (CPU tests), and -- because the data path needs no collective at all -- a plain
FILE group (`backend="none"`) that needs neither: if RCCL cannot initialise on a node (the nccl branch
had never run anywhere before round 3's scaling run), init() falls back to gloo and then to files
instead of losing the whole measurement."""
import json
import os
import sys
import tempfile
import time

BASE_SEED = 0xC0FFEE


def rank_env():
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0)))


def capture_seed(rank, streams_per_rank, s):
    """Seed of capture `s` of rank `rank`: globally unique, rank 0 reproduces the 1-GPU workload."""
    return BASE_SEED + rank * streams_per_rank + s


def device_of(file_index, n_devices):
    """File-per-GPU map of the batch CLI (`rtl_wmbus_hip -G all|list FILE...`, wm_main.c shard_slot) and of
    SURVEY.md 8(e): stream s -> position s mod n in the device list."""
    return file_index % n_devices


def owned_captures(rank, world, total):
    """Contiguous block partition of `total` captures (used when a fixed file list is sharded)."""
    per = (total + world - 1) // world
    return range(min(total, rank * per), min(total, (rank + 1) * per))


def host_threads_per_context(world, contexts, cpus=None):
    """Packet-decoder threads per receiver context such that ranks x contexts x threads stays within the host's
    hardware threads (8 ranks x 8 contexts x 8 threads = 512 on a 256-thread host was untried)."""
    cpus = cpus or os.cpu_count() or 16
    return max(1, min(16, cpus // max(1, world * contexts)))


class FileGroup:
    """Barrier and reductions through small files in a directory all ranks of the node can see (keyed by MASTER_PORT so
    that concurrent jobs do not meet).  O(world) files per operation; meant for a handful of scalars per run."""
    backend = "none"

    def __init__(self, rank, world, root=None, timeout_s=600.0, tag=""):
        self.rank, self.world, self.timeout_s, self.seq = rank, world, timeout_s, 0
        # one directory per launch: the launcher's nonce (bench.py's own spawner sets WMBUS_GROUP_TAG, torchrun its run id) or,
        # failing that, port + parent pid -- files a crashed run left behind are not met again
        nonce = os.environ.get('WMBUS_GROUP_TAG') or f"{os.environ.get('TORCHELASTIC_RUN_ID', '')}{os.getppid()}"
        key = f"wmbus_group{tag}_{os.environ.get('MASTER_PORT', '0')}_{nonce}"
        self.dir = os.path.join(root or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()), key)
        os.makedirs(self.dir, exist_ok=True)

    def _exchange(self, value):
        self.seq += 1
        mine = os.path.join(self.dir, f"{self.seq}.{self.rank}")
        with open(mine + ".tmp", "w") as f:
            json.dump(value, f)
        os.replace(mine + ".tmp", mine)                     # atomic: a reader never sees half a file
        out, t0 = [], time.monotonic()
        for r in range(self.world):
            p = os.path.join(self.dir, f"{self.seq}.{r}")
            while not os.path.exists(p):
                if time.monotonic() - t0 > self.timeout_s:
                    raise TimeoutError(f"rank {self.rank}: rank {r} did not reach step {self.seq} of the file group")
                time.sleep(0.002)
            out.append(json.load(open(p)))
        return out

    def barrier(self):
        self._exchange(0)

    def all_gather(self, value):
        return self._exchange(value)

    def destroy(self):
        self._exchange(0)
        bye = os.path.join(self.dir, f"bye.{self.rank}")      # "I have read everything I will ever read"
        open(bye, "w").close()
        if self.rank == 0:
            t0 = time.monotonic()
            while not all(os.path.exists(os.path.join(self.dir, f"bye.{r}")) for r in range(self.world)) and time.monotonic() - t0 < 60.0:
                time.sleep(0.002)
            for f in os.listdir(self.dir):
                try:
                    os.remove(os.path.join(self.dir, f))
                except OSError:
                    pass
            try:
                os.rmdir(self.dir)
            except OSError:
                pass


class TorchGroup:
    def __init__(self, dist, backend):
        self.dist, self.backend = dist, backend

    def barrier(self):
        import torch
        self.dist.barrier()
        if self.backend == "nccl":
            torch.cuda.synchronize()

    def all_gather(self, value):
        out = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(out, value)
        return out

    def destroy(self):
        self.dist.destroy_process_group()

    # what tests/gloo_worker.py and older callers use directly
    def __getattr__(self, name):
        return getattr(self.dist, name)


def init(world, local_rank, backend=None, rank=None, force=False):
    """Group for barrier / reductions only; None for a single rank.  backend: "nccl", "gloo", "none" (files) or None =
    nccl when torch sees a GPU, else gloo.  A backend that fails to initialise falls back to the next one: the
    measurement needs a barrier, not a particular transport."""
    if world <= 1 and not force:                            # force: a one-rank group (tests of the transport itself)
        return None
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    order = {"nccl": ["nccl", "gloo", "none"], "gloo": ["gloo", "none"], "none": ["none"]}
    if backend is None:
        try:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        except Exception:                                    # no torch at all: files
            backend = "none"
    # The ranks AGREE on the transport (decided rank by rank, one rank could sit in gloo while the others still
    # waited in RCCL's 120 s all-reduce, or a mixed torch / file world could form): after every attempt each rank casts its
    # result through a small file group -- which always works on one node -- and a backend is taken only if it came up on
    # every rank; otherwise all of them tear it down and try the next one together.
    votes = FileGroup(rank, world, tag="init") if world > 1 else None
    last = None
    for b in order.get(backend, [backend]):
        if b == "none":
            if votes is not None:
                votes.destroy()
            return FileGroup(rank, world)
        ok, grp = True, None
        try:
            import datetime
            import torch
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
            if b == "nccl":
                dev = local_rank % max(1, torch.cuda.device_count())     # a launcher may narrow the visible devices per rank
                torch.cuda.set_device(dev)
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=datetime.timedelta(seconds=120))
                t = torch.ones(1, device="cuda")
                dist.all_reduce(t)                           # RCCL builds its rings lazily: fail here, not inside the timed region
                torch.cuda.synchronize()
            else:
                dist.init_process_group(b, timeout=datetime.timedelta(seconds=300))
            grp = TorchGroup(dist, b)
        except Exception as e:                               # noqa: BLE001 -- any failure means "try the next transport"
            ok, last = False, e
            print(f"shard.init: backend {b} failed on rank {rank} ({e!r}); falling back", file=sys.stderr, flush=True)
        everyone = all(votes.all_gather(bool(ok))) if votes is not None else ok
        if everyone:
            if votes is not None:
                votes.destroy()
            return grp
        if ok:                                               # it came up here but not everywhere: leave it together
            try:
                grp.destroy()
            except Exception:                                # noqa: BLE001
                pass
            print(f"shard.init: backend {b} is up on rank {rank} but not on every rank; falling back with the others", file=sys.stderr, flush=True)
    raise RuntimeError(f"no process-group backend could be initialised: {last!r}")


def barrier(group):
    if group is not None:
        group.barrier()


def max_over_ranks(group, value):
    return value if group is None else max(float(v) for v in group.all_gather(float(value)))


def sum_over_ranks(group, value):
    return value if group is None else sum(float(v) for v in group.all_gather(float(value)))


def gather(group, value):
    """`value` (JSON-serialisable) of every rank, in rank order."""
    return [value] if group is None else group.all_gather(value)


def destroy(group):
    if group is not None:
        group.destroy()
