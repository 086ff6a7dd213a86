"""One-process-per-GPU helpers.  Inference shards images across ranks with NO data-path collective;
the E2E-FT training step exchanges UNet gradients once per optimizer step (the reference does this through
accelerate/DDP: training/train.py:369,563; training/scripts/multi_gpu.yaml:1-15) — that exchange lives in training.FlatAdamW (slices of
one flat gradient buffer, all-reduced from autograd hooks), the only gradient exchange in the product.  Backend "nccl" is RCCL on ROCm;
the same code runs on gloo for the CPU tests."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* when WORLD_SIZE > 1. Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (step time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_mean(loss, repeat=1, group=None):
    """`accelerator.gather(loss.repeat(train_batch_size)).mean()` (training/train.py:559, GeoWizard train_depth_normal.py:772): the logging value of a
    micro-step = the mean over ranks of each rank's scalar loss (the repeat only weights every rank by its batch size, equal on all ranks).  One all-gather of
    `repeat` floats per rank; returns a 0-d tensor on the loss's device WITHOUT synchronising the host (the reference calls `.item()` on it every micro-step;
    accumulate on the device and read back when you log)."""
    t = loss.detach().reshape(1).repeat(int(repeat)).contiguous()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t.mean()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return torch.cat(out).mean()
