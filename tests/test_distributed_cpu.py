"""world_size-2 gloo test of the batch sharding used by bench.py / ShardedEvaluator (CPU only; the
per-rank compute is the oracle standing in for the per-GPU engine)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank, world, port, B, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import encrypt_bits, keygen, mk, oracle_scheme
    from mktfhe_amd import distributed as D
    r, w, _ = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    p = mk.KMS2party.scaled(n=8, N=64)
    crs, keys = keygen(p, 51)           # same seed on every rank = replicated keys
    so = oracle_scheme(p, crs, keys)
    bits = (np.arange(2 * B) * 7 % 3 == 0)
    c = encrypt_bits(p, keys, bits, seed=5100)
    ev = D.ShardedEvaluator(lambda op, x, y: so.gate_batch(op, x, y, threads=1))
    full = ev.gate(0, c[:B], c[B:], gather=True)
    tmax = D.max_over_ranks(1.0 + rank)
    if rank == 0:
        ref = so.gate_batch(0, c[:B], c[B:], threads=1)
        ret["ok"] = bool(np.array_equal(full, ref))
        ret["tmax"] = tmax
        ret["dec"] = bool(np.array_equal(mk.lwe_decrypt(full, keys, p), ~(bits[:B] & bits[B:])))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("B", [6, 7])      # even and ragged split
def test_sharded_gates_gloo_world2(B):
    world = 2
    port = 29600 + B + (os.getpid() % 200)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, ret)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(180)
        assert pr.exitcode == 0
    assert ret["ok"] and ret["dec"] and ret["tmax"] == 2.0


def _fallback_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from mktfhe_amd import distributed as D
    D.init_process_group("nccl", device=None)          # no GPU here: RCCL cannot start on any rank -> gloo
    t = torch.ones(1)
    torch.distributed.all_reduce(t)
    if rank == 0:
        ret["backend"], ret["sum"] = D.ACTIVE_BACKEND, float(t.item())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_rendezvous_falls_back_to_gloo_when_rccl_cannot_start():
    """bench.py's ranks rendezvous over RCCL; where RCCL cannot be created (here: no GPU at all; on a node: no peer access, a
    refused IPC mode) every rank falls back to gloo instead of losing the run -- the data path has no collective"""
    world, port = 2, 29900 + (os.getpid() % 90)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, ret)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(180)
        assert pr.exitcode == 0
    assert ret["backend"] == "gloo" and ret["sum"] == 2.0


def _lone_failure_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from mktfhe_amd import distributed as D
    D.init_process_group("gloo", device=None, fallback="gloo", agree_s=60, _test_fail_ranks=(1,))     # rank 1's probe fails locally, rank 0's succeeds
    t = torch.ones(1)
    torch.distributed.all_reduce(t)
    ret[rank] = (D.ACTIVE_BACKEND, float(t.item()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_ranks_agree_before_switching_backend_when_only_one_of_them_failed():
    """one rank's probe of the first process group fails locally, the other's succeeds: the verdicts are exchanged first, so BOTH move to
    the fallback group on the verdict store's fresh key space (a rank that switched alone would leave the other waiting in a collective until its timeout)"""
    world, port = 2, 29700 + (os.getpid() % 90)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_lone_failure_worker, args=(r, world, port, ret)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(180)
        assert pr.exitcode == 0
    assert ret[0] == ret[1] == ("gloo", 2.0)


def test_fallback_under_the_drivers_launcher(tmp_path):
    """the driver's launch line (python -m torch.distributed.run --master-addr 127.0.0.1 --master-port P): the ranks are CLIENTS of the launcher's
    store, so the fallback group cannot rendezvous on a port of its own -- it is built on the verdict store rank 0 hosts (no GPU here: RCCL
    fails on every rank, both fall back to gloo and reduce)"""
    import subprocess
    w = tmp_path / "w.py"
    w.write_text(f"""import os, sys, torch
sys.path.insert(0, {ROOT!r})
from mktfhe_amd import distributed as D
r, w, l = D.init_process_group("nccl", device=None, agree_s=60, timeout_s=120)
t = torch.ones(1); torch.distributed.all_reduce(t)
print("RANK", r, D.ACTIVE_BACKEND, float(t.item()), flush=True)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
""")
    port = 29800 + (os.getpid() % 90)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(w)], capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RANK 0 gloo 2.0" in out.stdout and "RANK 1 gloo 2.0" in out.stdout, out.stdout


def test_shard_slices():
    from mktfhe_amd.distributed import shard_slices
    assert shard_slices(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_slices(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert shard_slices(0, 2) == [(0, 0), (0, 0)]


def test_bench_gpus_n_spawns_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` without a torchrun environment starts two rank processes itself.  On this CPU-only
    box each rank dies at torch.cuda.set_device: the parent must exit non-zero and print no JSON line (never a silent
    one-rank run reported as n_gpus 1)."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""; env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4",
                        "--no-cpu-baseline", "--no-roofline", "--no-secondary"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "ranks failed" in r.stderr
