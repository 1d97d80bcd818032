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


def init_from_env(backend=None, force=None, timeout_s=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, world, local_rank).

    ``force`` (default: PFO_DIST_FORCE=1 in the environment): initialise the process group at world 1 too, so that the
    rank path - RCCL communicator, the all-reduce of the flat gradient - runs end to end on a single GPU.
    The group gets an explicit timeout (default 180 s, PFO_DIST_TIMEOUT_S): a rank that never arrives at the rendezvous or
    at a collective fails the job quickly instead of holding it for the backend's 10-30 minute default.  With RCCL the
    communicator is bound to this rank's device at once (``device_id``): no lazy initialisation inside the first timed step."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force is None:
        force = os.environ.get("PFO_DIST_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # PFO_DIST_BACKEND is a test hook (2 ranks sharing one GPU over gloo); production = nccl (RCCL)
            backend = os.environ.get("PFO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if timeout_s is None:
            timeout_s = float(os.environ.get("PFO_DIST_TIMEOUT_S", "180"))
        kw = dict(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
        if backend == "nccl":
            dev = int(os.environ["PFO_FORCE_DEVICE"]) if os.environ.get("PFO_FORCE_DEVICE") is not None else local
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(**kw)
    return rank, world, local


def shard_bounds(batch, rank, world):
    """Contiguous shard [lo, hi) of a batch of ``batch`` interactions owned by ``rank``: balanced (sizes differ by at
    most one); shards are empty only when the batch is shorter than the world size (TGN.embed_device handles that)."""
    return rank * batch // world, (rank + 1) * batch // world


def allreduce_flat_grad(flat_grad, world, force=False):
    """The step's single collective: sum of the flat gradient buffer over all ranks (``force``: also at world 1)."""
    if world > 1 or force:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


_COMM_STREAMS = {}


def allreduce_flat_grad_buckets(tgn, world, force=False):
    """The same sum in TWO pieces, the first one beside the backward.  With ``tgn.dp_bucketed`` set, the native backward
    records an event when the top layer's gradient block ``flat_grad[tgn.grad_split:]`` is final - layers L-1 .. 1 and the
    GRU are still being differentiated then (roughly the second half of the backward at L = 2).  A communication stream
    waits for that event and reduces the block; the rest is reduced on the caller's stream after the backward; the
    caller's stream then waits for the communication stream.  Element-wise sums do not depend on how the buffer is cut, so
    the result is bit-identical to ``allreduce_flat_grad``.  ``force``: run the collectives at world 1 too (tests)."""
    g = tgn.flat_grad
    if g is None or (world <= 1 and not force):
        return g
    split = tgn.grad_split
    # The NUMBER and the SIZES of the collectives must be the same on every rank: the cut is chosen from rank-invariant
    # state only (the model's layout and the dp_bucketed switch).  A rank whose backward did not run this step (empty
    # shard: batch shorter than the world) has no fresh event - its zero gradient is final already, so it issues the same
    # two collectives in the same order and only skips the wait.
    if not (tgn.dp_bucketed and 0 < split < g.numel()):
        dist.all_reduce(g, op=dist.ReduceOp.SUM)            # one layer / bucketing off: one piece, on every rank
        return g
    fresh, tgn._bucket_event_fresh = tgn._bucket_event_fresh, False
    if g.device.type != "cuda":                             # host tensors (gloo rehearsal of the call pattern): no streams
        dist.all_reduce(g[split:], op=dist.ReduceOp.SUM)
        dist.all_reduce(g[:split], op=dist.ReduceOp.SUM)
        return g
    main = torch.cuda.current_stream(g.device)
    side = _COMM_STREAMS.get(g.device)
    if side is None:
        side = _COMM_STREAMS[g.device] = torch.cuda.Stream(device=g.device)
    if fresh:
        side.wait_event(tgn._bucket_event)
    else:
        side.wait_stream(main)                              # whatever produced (or cleared) the buffer on the caller's stream
    with torch.cuda.stream(side):
        dist.all_reduce(g[split:], op=dist.ReduceOp.SUM)
    dist.all_reduce(g[:split], op=dist.ReduceOp.SUM)
    main.wait_stream(side)
    return g


def allreduce_flat_grad_ordered(tgn, world, force=False):
    """The two pieces of ``allreduce_flat_grad_buckets`` for a FUSED step whose optimizer runs per bucket in order of first use
    (``tgn.dp_ordered``; ``FusedAdam.step(side=True)``).  Called on the library's side stream (``bpr_step``): the top layer's
    block is reduced on the communication stream as soon as its gradients are final - beside the rest of the backward - and
    this stream reduces ``flat_grad[:split]`` (time encoder, GRU, layer 1: what the next forward reads FIRST) behind the
    backward's end.  Unlike the two-bucket form it does not join the communication stream: the optimizer steps the first-use
    block at once, the next forward waits for that kernel alone, and the top block's step follows behind
    ``tgn.wait_comm_stream()``.  Same collectives, same sizes, same order on every rank (also one with an empty shard)."""
    g = tgn.flat_grad
    tgn._comm_pending = None
    if g is None or (world <= 1 and not force):
        return g
    split = tgn.grad_split
    if not (tgn.dp_bucketed and 0 < split < g.numel()):
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g
    fresh, tgn._bucket_event_fresh = tgn._bucket_event_fresh, False
    if g.device.type != "cuda":
        dist.all_reduce(g[split:], op=dist.ReduceOp.SUM)
        dist.all_reduce(g[:split], op=dist.ReduceOp.SUM)
        return g
    main = torch.cuda.current_stream(g.device)
    side = _COMM_STREAMS.get(g.device)
    if side is None:
        side = _COMM_STREAMS[g.device] = torch.cuda.Stream(device=g.device)
    if fresh:
        side.wait_event(tgn._bucket_event)
    else:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        dist.all_reduce(g[split:], op=dist.ReduceOp.SUM)
    dist.all_reduce(g[:split], op=dist.ReduceOp.SUM)
    tgn._comm_pending = side                               # joined by the optimizer in front of the top block's step
    return g


def broadcast_parameters(flat_params, world, src=0):
    if world > 1:
        dist.broadcast(flat_params, src=src)
    return flat_params
