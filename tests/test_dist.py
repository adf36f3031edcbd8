"""world_size-2 checks (gloo, CPU) of the data-parallel harness that bench.py uses on RCCL: rank
discovery, per-rank shards and seeds, barrier, max-over-ranks time and whole-job throughput.  The op
itself needs no collective, so this is everything that crosses ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from uvhand_amd import harness
    import torch.distributed as dist
    assert harness.dist_env() == (rank, rank, world)
    assert harness.init_process_group("gloo")
    lo, hi = harness.shard_batch(5, rank, world)
    seed = harness.rank_seed(1000, rank)
    # every rank synthesises its own shard; rank r pretends to take (r + 1) * 0.25 s for `hi - lo` samples
    elapsed = 0.25 * (rank + 1)
    harness.barrier()
    slowest = harness.max_over_ranks(elapsed)
    total = harness.sum_over_ranks(hi - lo)
    thr = harness.job_throughput(hi - lo, elapsed)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, seed))
    torch.save(dict(slowest=slowest, total=total, thr=thr, gathered=gathered), os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_harness_on_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / ("r%d.pt" % r), weights_only=False) for r in range(world)]
    for r in res:
        assert r["slowest"] == pytest.approx(0.5)            # max over ranks, identical everywhere
        assert r["total"] == 5                                # shards cover the global batch once
        assert r["thr"] == pytest.approx(5 / 0.5)
    shards = res[0]["gathered"]
    assert [(lo, hi) for lo, hi, _ in shards] == [(0, 3), (3, 5)]
    assert len({seed for _, _, seed in shards}) == world      # distinct seeds per rank


def test_single_process_defaults(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    sys.path.insert(0, ROOT)
    from uvhand_amd import harness
    assert harness.dist_env() == (0, 0, 1)
    assert harness.init_process_group("gloo") is False
    assert harness.shard_batch(7, 0, 1) == (0, 7)
    assert harness.max_over_ranks(1.5) == 1.5 and harness.job_throughput(6, 2.0) == 3.0
    covered = []
    for r in range(4):
        lo, hi = harness.shard_batch(10, r, 4)
        covered += list(range(lo, hi))
    assert covered == list(range(10))


def test_two_ranks_on_one_device_are_a_launch_error_under_rccl_only():
    """harness.duplicate_devices / check_one_rank_per_device: under backend "nccl" (RCCL) two ranks on one physical GPU must
    fail the job; the gloo rehearsal shares a card on purpose."""
    from uvhand_amd import harness
    a = {"host": "node0", "device": "cuda:0", "index": 0, "name": "x", "pci_bus_id": "0000:05:00", "uuid": "GPU-1"}
    b = dict(a, device="cuda:1", index=1, pci_bus_id="0000:15:00", uuid="GPU-2")
    assert harness.duplicate_devices([a, b]) == []
    assert harness.duplicate_devices([a, dict(a)]) == [(0, 1, "GPU-1 on node0")]
    assert harness.duplicate_devices([dict(a, uuid=None), dict(a, uuid=None, device="cuda:0")]) == [(0, 1, "0000:05:00 on node0")]
    # two NODES: the same bus id / index / a constant uuid on another host is another GPU
    assert harness.duplicate_devices([a, dict(a, host="node1")]) == []
    assert harness.duplicate_devices([dict(a, uuid=None, pci_bus_id=None), dict(a, uuid=None, pci_bus_id=None, host="node1")]) == []
    assert "host" in harness.device_identity(None)
    cpu = harness.device_identity(None)
    assert harness.duplicate_devices([cpu, cpu]) == []                  # nothing to tell apart on the CPU
    assert harness.check_one_rank_per_device("nccl", None) == [cpu]     # one process: nothing to check
    info = harness.collective_runtime()
    assert "rccl" in info and "visible_devices" in info
