"""Small host-side helpers for running the op data-parallel, one process per GPU.

The op shards over the batch with no exchange (every (b, q, m) output depends only on value[b],
loc[b, q, m], attn[b, q, m]; grad_value[b] only receives from queries of b — reference kernel
ms_deform_im2col_cuda.cuh:255-297), which is how the reference trains: DistributedSampler gives each
rank whole samples (datasets/samplers.py:16) and DDP all-reduces parameter gradients (main.py:96-98).
These helpers hold what `bench.py` and the tests share: rank discovery from the torchrun
environment (tools/launch.py:158-187 sets the same variables), per-rank seeds / batch slices, and the
max-over-ranks timing reduction.  They work on any torch.distributed backend (RCCL on the GPUs, gloo in
the CPU tests) and contain no compute.
"""
import os

import socket

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) from the launcher's environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend, device=None):
    """env:// rendezvous on 127.0.0.1 unless the launcher says otherwise; no-op for one process."""
    rank, _, world = dist_env()
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kwargs = {}
    if device is not None and backend == "nccl":
        kwargs["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return True


def rank_seed(base_seed, rank):
    """Distinct, reproducible input seed per rank (each rank synthesises its own batch shard)."""
    return int(base_seed) + 1000003 * int(rank)


def shard_batch(global_batch, rank, world):
    """Contiguous slice [lo, hi) of a global batch owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(int(global_batch), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier(device=None):
    """All ranks' queued GPU work done, every rank arrived, and (NCCL's barrier is itself stream work) that
    arrival observed by the host: synchronize, barrier, synchronize."""
    cuda = device is not None and device.type == "cuda"
    if cuda:
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        if cuda:
            torch.cuda.synchronize(device)


def max_over_ranks(seconds, device=None):
    """Slowest rank's elapsed time (what the whole job waited for)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def job_throughput(samples_this_rank, seconds_this_rank, device=None):
    """Whole-job samples/s: all ranks' samples over the slowest rank's time."""
    return sum_over_ranks(samples_this_rank, device) / max_over_ranks(seconds_this_rank, device)


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (a one-element list without a process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def device_identity(device):
    """What tells two ranks' GPUs apart: index, name, PCI bus id and uuid where this torch exposes them."""
    host = socket.gethostname()
    if device is None or device.type != "cuda":
        return {"host": host, "device": str(device), "index": None, "name": None, "pci_bus_id": None, "uuid": None}
    props = torch.cuda.get_device_properties(device)
    bus = getattr(props, "pci_bus_id", None)
    return {"host": host, "device": str(device), "index": device.index, "name": props.name,
            "pci_bus_id": None if bus is None else "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), bus,
                                                                       getattr(props, "pci_device_id", 0)),
            "uuid": str(getattr(props, "uuid", "")) or None}


def duplicate_devices(identities):
    """Pairs of ranks that sit on the same physical GPU: same HOST and same uuid / PCI bus id, or — lacking both — same
    index.  (Bus ids and indices repeat from node to node, and some builds report a constant uuid: without the host a
    multi-node job would see every node's GPU 0 as one device.)"""
    seen, dup = {}, []
    for rank, ident in enumerate(identities):
        dev = ident.get("uuid") or ident.get("pci_bus_id") or ident.get("index")
        if dev is None:
            continue
        key = (ident.get("host"), dev)
        if key in seen:
            dup.append((seen[key], rank, "%s on %s" % (dev, ident.get("host"))))
        else:
            seen[key] = rank
    return dup


def collective_runtime():
    """Version of the collective library behind backend "nccl" on this build (RCCL on ROCm), device count, HIP version."""
    info = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None), "visible_devices": torch.cuda.device_count()}
    try:
        info["rccl"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as exc:                                # no GPU build / no collective library
        info["rccl"] = "unavailable (%s)" % type(exc).__name__
    return info


def check_one_rank_per_device(backend, device):
    """Every rank's device identity, gathered; under RCCL two ranks on one GPU is a launch error (it deadlocks or silently
    serialises): raise on EVERY rank so that the job exits non-zero instead of reporting a number.  Under gloo (the CPU /
    one-card rehearsal) sharing a device is the point, so it is only reported.  Returns the list of identities."""
    idents = gather_objects(device_identity(device))
    dup = duplicate_devices(idents)
    if dup and backend == "nccl":
        raise RuntimeError("ranks share a GPU under backend nccl (RCCL needs one rank per device): %s — launch with one "
                           "process per GPU (torch.distributed.run --nproc-per-node <#GPUs>) and LOCAL_RANK < device_count (%d)"
                           % (", ".join("ranks %d and %d on %s" % d for d in dup), torch.cuda.device_count()))
    return idents
