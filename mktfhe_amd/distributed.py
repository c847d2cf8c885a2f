"""Multi-GPU sharding of independent gate bootstraps: one process per GPU, keys replicated per GPU,
contiguous slices of the batch per rank, NO data-path collective.  torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" on CPU for tests) is used only for rendezvous, the timing
barrier / max-over-ranks, and -- when a caller wants the full result everywhere -- one all_gather.

The reference has no distributed layer (Base.Threads only: bootstrapping.jl:376-378, :343, :573); gates
in a batch are independent and the scheme object is read-only during evaluation, which is what makes
the batch dimension shardable."""
import os

import numpy as np


def env():
    """(rank, world_size, local_rank) from the torch.distributed.run environment"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_slices(B, world):
    """contiguous, balanced slices of range(B): rank r gets [start, stop)"""
    base, rem = divmod(B, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


ACTIVE_BACKEND = None      # what init_process_group ended up with ("nccl" = RCCL on ROCm, or "gloo")
_VERDICTS = None           # the agreement store stays alive for the life of the process


def _agree(store, rank, world, stage, ok, wait_s):
    """every rank publishes whether `stage` worked for it and reads every other rank's verdict: the ranks switch backend together or
    not at all (a rank that fell back alone would leave the others waiting in RCCL until their collective timeout)"""
    import datetime
    store.set(f"{stage}/{rank}", "1" if ok else "0")
    keys = [f"{stage}/{r}" for r in range(world)]
    store.wait(keys, datetime.timedelta(seconds=wait_s))          # raises after wait_s: a loud error instead of a 900 s hang
    return all(store.get(k) == b"1" for k in keys)


def _verdict_store(rank, world, port, wait_s):
    """The agreement store: a TCPStore of its own on MASTER_ADDR : MASTER_PORT + 2, hosted by rank 0 and created for THIS launch -- it works across
    nodes (the file store of round 5 did not) and cannot hold a previous launch's verdicts; the keys additionally carry the elastic launcher's run id and
    restart count where there is one.  A rank that cannot reach it raises: acting alone is exactly what the store exists to prevent."""
    import datetime
    import torch.distributed as dist
    store = dist.TCPStore(os.environ["MASTER_ADDR"], port + 2, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=wait_s), wait_for_workers=False)
    nonce = os.environ.get("TORCHELASTIC_RUN_ID", "run") + "." + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    return dist.PrefixStore(f"mkt/{nonce}", store), dist.PrefixStore(f"mkt/{nonce}/fallback_group", store)


def init_process_group(backend=None, device=None, fallback="gloo", timeout_s=900, agree_s=180, _test_fail_ranks=()):
    """Rendezvous of the ranks (nothing on the data path: gates shard with no collective).  backend "nccl" is RCCL; it is
    PROVED with one tiny all-reduce right away, and if creating or proving it fails on ANY rank -- no peer access between the
    visible devices, an IPC mode the host driver refuses -- ALL ranks fall back to `fallback` (gloo over TCP, CPU tensors)
    instead of losing the run: the barrier, the max-over-ranks and the census are all it carries.  The outcome of each stage
    (communicator created; probe all-reduce correct) is agreed through a small store of its own (_verdict_store) before any rank acts on
    it, and nobody enters the probe collective unless every rank holds a communicator.  What this cannot shorten is a rank that never
    ARRIVES at a collective (creation and the probe are collectives themselves): the others then wait out `timeout_s`.
    Ports: MASTER_PORT (the group), + 2 (the verdicts AND the fallback group's rendezvous: under torch.distributed.run the ranks are clients of the
    launcher's store on MASTER_PORT, so a second env:// rendezvous on another port would find no server -- the fallback group is built on the
    verdict store, which rank 0 hosts itself).  `_test_fail_ranks`: tests only -- these ranks report a local failure of the probe after its collective."""
    import datetime
    import sys
    import torch
    import torch.distributed as dist
    global ACTIVE_BACKEND, _VERDICTS
    rank, world, local = env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or "nccl"
        to = datetime.timedelta(seconds=timeout_s)
        port = int(os.environ["MASTER_PORT"])
        verdicts = fb_store = None
        if fallback:                                       # (fallback == backend: one retry of the same backend on a fresh store)
            verdicts, fb_store = _VERDICTS = _verdict_store(rank, world, port, agree_s)     # raises if unreachable: fatal on that rank, and the others time out on its verdict
        why = None
        try:
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=to, **kw)
        except Exception as e:      # noqa: BLE001
            if verdicts is None:
                raise
            why = e
        ok = why is None if verdicts is None else _agree(verdicts, rank, world, "created", why is None, agree_s)
        if ok and (backend == "nccl" or _test_fail_ranks):
            try:
                probe = torch.ones(1, device=(device if device is not None else "cuda") if backend == "nccl" else "cpu")
                dist.all_reduce(probe)
                if backend == "nccl":
                    torch.cuda.synchronize()
                assert int(probe.item()) == world
                if rank in _test_fail_ranks:
                    raise RuntimeError("simulated local failure on this rank only")
            except Exception as e:  # noqa: BLE001
                if verdicts is None:
                    raise
                why = e
            if verdicts is not None:
                ok = _agree(verdicts, rank, world, "proved", why is None, agree_s)
        if not ok:
            sys.stderr.write(f"mktfhe_amd.distributed: rank {rank}: backend {backend} unusable on at least one rank"
                             f"{'' if why is None else f' (here: {type(why).__name__}: {str(why)[:200]})'}; all ranks fall back to {fallback}\n")
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:       # noqa: BLE001
                pass
            backend = fallback
            dist.init_process_group(backend, store=fb_store, rank=rank, world_size=world, timeout=to)     # a fresh key space: every rank takes the same step
        ACTIVE_BACKEND = backend
        if verdicts is not None:
            dist.barrier()                                 # every rank has read every verdict; the store lives as long as rank 0's process
    return rank, world, local


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (the bench's step time)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class ShardedEvaluator:
    """Strong-scaling helper: every rank holds the same full batch description, evaluates its own
    slice with `gate_fn(op, x, y) -> out` (the per-GPU Scheme.gate) and, if asked, all-gathers the
    result.  No collective sits between input and output of a gate."""

    def __init__(self, gate_fn, rank=None, world=None):
        r, w, _ = env()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.gate_fn = gate_fn

    def gate(self, op, x, y, gather=True):
        import torch
        import torch.distributed as dist
        B = x.shape[0]
        s, e = shard_slices(B, self.world)[self.rank]
        mine = self.gate_fn(op, x[s:e], y[s:e])
        if not gather or self.world == 1:
            return mine
        is_np = isinstance(mine, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(mine).view(np.int32)) if is_np else mine
        sizes = [b - a for a, b in shard_slices(B, self.world)]
        parts = [torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for n in sizes]
        dist.all_gather(parts, t) if len(set(sizes)) == 1 else _all_gather_ragged(parts, t, self.rank)
        full = torch.cat(parts, 0)
        return full.numpy().view(np.uint32) if is_np else full


def _all_gather_ragged(parts, t, rank):
    import torch.distributed as dist
    for r, buf in enumerate(parts):
        if r == rank:
            buf.copy_(t)
        dist.broadcast(buf, src=r)
