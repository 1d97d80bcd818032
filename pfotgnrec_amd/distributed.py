"""Edge-batch data parallelism over RCCL.

One process per GPU.  Every rank holds the full replicated state (parameters, Adam moments, CSR,
features, memory, pending-message table); a global batch of B interactions is cut into ``world``
contiguous shards and each rank samples/embeds/back-propagates only its shard's roots.  Exactly one
collective per step: an all-reduce (sum) of the flat fp32 gradient buffer, with the local mean-loss
gradient pre-scaled by (local batch / global batch) so the sum is the global-batch mean gradient.  The memory persist and
raw-message store run for ALL global positives on every rank (they depend only on replicated state),
so replicas stay identical without a second exchange.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # PFO_DIST_BACKEND is a test hook (2 ranks sharing one GPU over gloo); production = nccl (RCCL)
            backend = os.environ.get("PFO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(batch, rank, world):
    """Contiguous shard [lo, hi) of a batch of ``batch`` interactions owned by ``rank``: balanced (sizes differ by at
    most one); shards are empty only when the batch is shorter than the world size (TGN.embed_device handles that)."""
    return rank * batch // world, (rank + 1) * batch // world


def allreduce_flat_grad(flat_grad, world):
    """The step's single collective: sum of the flat gradient buffer over all ranks."""
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def broadcast_parameters(flat_params, world, src=0):
    if world > 1:
        dist.broadcast(flat_params, src=src)
    return flat_params
