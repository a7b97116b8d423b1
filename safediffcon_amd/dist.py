"""Batch-of-trajectories sharding: trajectories are independent given the weights, so the
batch axis is split in contiguous blocks over one process per GPU; weights are replicated; no collective runs
during the 1000 denoising steps.  The only exchange is the conformal all-gather (safediffcon_amd.conformal).

Noise: every sample() call draws its Philox key from torch's global CPU generator and mixes the process rank into it
(diffusion._mix_rank), so ranks that all call ``torch.manual_seed(cfg.seed)`` -- the usual launcher pattern -- still
sample their shards from different noise streams (rank 0 and a single-process run use the plain key)."""
import os

import torch


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n, rank=None, world_size=None):
    """contiguous, equal-size block of [0, n) for this rank (n must divide evenly: the all-gather is unpadded)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    if n % world_size:
        raise ValueError(f"batch {n} does not split evenly over {world_size} ranks")
    per = n // world_size
    return rank * per, (rank + 1) * per


def shard(t, dim=0):
    lo, hi = shard_range(t.shape[dim])
    return t.narrow(dim, lo, hi - lo)


def init_from_env(backend=None):
    """one process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  The rank binds
    its device FIRST and hands it to init_process_group (`device_id`): the RCCL communicator is then created eagerly on
    that device instead of lazily on whatever device the first collective happens to see."""
    import torch.distributed as dist
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local)
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if use_gpu else "gloo")     # "nccl" IS RCCL on ROCm
        kw = dict(device_id=torch.device("cuda", local)) if (backend == "nccl" and use_gpu) else {}
        dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=ws, **kw)
    return int(os.environ.get("RANK", "0")), ws, local
