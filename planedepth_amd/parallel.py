"""Data-parallel plumbing for the hot path.

The warp+loss path shards over batch elements with NO data-path collective: every image (and target view) is
independent, and the only cross-sample coupling is the final ``.mean()`` which each rank takes over its own shard —
exactly what DDP expects (the reference never all-reduces the loss; trainer.py:52,139-142,742).  So multi-GPU here is:
one process per GPU, ``DistributedSampler``-style contiguous shards, and collectives only for (a) the timing barrier
and max-over-ranks reduction of ``bench.py`` and (b) the optional reporting of a global loss.  Gradients of the
*networks* are all-reduced by stock DDP over RCCL (xGMI) — outside this package.
"""
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank).

    backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" for CPU tests; default picks by CUDA availability.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # a rank that died must not leave the others (and the GPU box) waiting in a collective for the default 10-30 min
        import datetime
        timeout = datetime.timedelta(seconds=int(os.environ.get("PD_PG_TIMEOUT_S", "240")))
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"),
                                rank=rank, world_size=world, timeout=timeout)
    return rank, world, local_rank


def shard_range(global_batch, rank, world):
    """Contiguous shard [lo, hi) of a global batch, like trainer.py:52 (batch_size // n_gpu per rank)."""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def shard_batch(tensors, rank, world, global_batch):
    """Slice every [global_batch, ...] tensor of a dict to this rank's shard; anything else (scalars, tensors that
    broadcast over the batch) passes through."""
    lo, hi = shard_range(global_batch, rank, world)
    out = {}
    for k, v in tensors.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == global_batch:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def barrier(device=None):
    """Device sync + process barrier (+ device sync again), the bracket bench.py puts around its timed region."""
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
        if device is not None and torch.cuda.is_available():
            torch.cuda.synchronize(device)


def _comm_device(device):
    """NCCL/RCCL reduces device tensors, gloo host tensors."""
    return device if (device is not None and dist.get_backend() == "nccl") else "cpu"


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the slowest rank defines the step time)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def global_mean(local_mean, device=None):
    """Mean over ranks of per-rank means (equal shard sizes) — reporting only; gradients never need it."""
    if not dist.is_initialized():
        return float(local_mean)
    t = torch.tensor([float(local_mean)], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()


def throughput(images_per_rank_per_step, steps, world, elapsed_max):
    """Whole-job images/sec: every rank's images over the slowest rank's time."""
    return images_per_rank_per_step * world * steps / elapsed_max
