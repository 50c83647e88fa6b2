"""Multi-GPU plumbing shared by bench.py and the CPU (gloo) tests: one process per GPU, reads sharded with no data-path
collective, ONE all-reduce of the per-position coverage track at end of run (the reference's MPI Allreduce,
src/Driver.cpp:1660-1672) and a max-over-ranks of the step time.  torch.distributed only — backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" on CPU."""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend, device=None):
    rank, world, _ = env_rank()
    # GM_FORCE_DIST=1: initialise the process group for a single rank too (rehearses the RCCL path on a one-GPU box)
    if (world > 1 or os.environ.get("GM_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime
        tmo = datetime.timedelta(minutes=60)          # a rank may spend minutes loading a human-size index replica
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    return rank, world


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def shard_range(n_items, rank, world):
    """contiguous block of items for this rank (the reference's MPI mode skips rank * (N / nproc + 1) reads, SeqManager.h:333-341)"""
    per = n_items // world + (1 if n_items % world else 0)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def read_seed(base_seed, rank):
    """per-rank seed of the synthetic read generator (weak scaling: every rank draws its own reads)"""
    return int(base_seed) + 1000 * int(rank)


def max_over_ranks(seconds, device="cpu"):
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_coverage(track):
    """IN-PLACE SUM all-reduce of a float32 coverage track: a CPU tensor under gloo, under nccl (= RCCL over xGMI) the
    HBM-resident track of the library itself (DeviceTrack view of gm_coverage_device_ptr(); no staging copy: the
    collective reads and writes the library's buffer)."""
    assert track.dtype == torch.float32 and track.is_contiguous()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(track, op=dist.ReduceOp.SUM)
    return track


class DeviceTrack:
    """zero-copy torch view of gm_coverage_device_ptr() (float[bins] in HBM) for the RCCL all-reduce"""

    def __init__(self, ptr, bins):
        self.__cuda_array_interface__ = {"shape": (int(bins),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}

    def tensor(self, device):
        return torch.as_tensor(self, device=device)
