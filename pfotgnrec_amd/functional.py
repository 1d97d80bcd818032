"""Autograd-facing wrappers of the small native ops (BPR loss, TimeEncode)."""
import torch

from . import _lib


_TICKETS = {}      # device -> i32[1]: the loss kernel's "last workgroup takes the mean" counter (zero at rest)


def bpr_loss_and_grad(emb, B, pos_off, neg_off, n_neg, scale):
    """One launch (``pfo_bpr_loss_fused``): (loss f32[1], d loss / d emb * scale f32[R,D])."""
    _lib.require_gpu(emb.device)
    emb = emb.contiguous()
    R, D = emb.shape
    ticket = _TICKETS.get(emb.device)
    if ticket is None:
        ticket = _TICKETS[emb.device] = torch.zeros(1, dtype=torch.int32, device=emb.device)
    loss = torch.empty(1, dtype=torch.float32, device=emb.device)
    d_emb = torch.empty_like(emb)
    scratch = torch.empty(B, dtype=torch.float32, device=emb.device)
    _lib.call("pfo_bpr_loss_fused", emb.data_ptr(), B, D, pos_off, neg_off, n_neg, R, float(scale), loss.data_ptr(),
              d_emb.data_ptr(), scratch.data_ptr(), ticket.data_ptr(), _lib.stream_ptr())
    return loss, d_emb


class _BprFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, B, pos_off, neg_off, n_neg, scale):
        loss, d_emb = bpr_loss_and_grad(emb, B, pos_off, neg_off, n_neg, scale)
        ctx.save_for_backward(d_emb)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d_emb,) = ctx.saved_tensors
        return d_emb * g, None, None, None, None, None


def bpr_loss(emb, batch, n_neg, pos_block=1, grad_scale=1.0):
    """BPR loss of main.py:321-337 / 364-381 on the root-ordered embedding matrix of ``TGN.embed_device``.

    emb [R,D] = [src B | dst B | (p_pos B) | neg B*n_neg]; ``pos_block`` = 1 when the positive is the destination
    (baseline path), 2 when it is the p_pos block (``ours`` path).  ``grad_scale`` multiplies the gradient only
    (``tgn.dp_grad_scale`` = local/global batch under data parallelism, so that the summed gradients equal the
    global-batch mean gradient also when the shards are uneven).
    """
    if batch == 0:
        # an empty data-parallel shard (global batch shorter than the world size): zero loss, zero gradient, but still
        # a differentiable scalar so that every rank runs the same backward / all-reduce sequence
        return emb.sum() * 0.0
    pos_off = pos_block * batch
    neg_off = (pos_block + 1) * batch
    return _BprFn.apply(emb, batch, pos_off, neg_off, n_neg, grad_scale)


def adjacent_rows(blocks):
    """Row blocks that lie one behind the other in ONE buffer (the outputs of ``compute_temporal_embeddings``, the gradient
    blocks ``bpr_loss_blocks`` hands back) as the single matrix they are a cut of - no copy; None when they are anything else."""
    t0 = blocks[0]
    if t0.dim() != 2:
        return None
    D, off, base = t0.shape[1], t0.storage_offset(), t0.untyped_storage().data_ptr()
    rows = 0
    for t in blocks:
        if (t.dim() != 2 or t.shape[1] != D or t.dtype != t0.dtype or not t.is_contiguous()
                or t.untyped_storage().data_ptr() != base or t.storage_offset() != off + rows * D):
            return None
        rows += t.shape[0]
    return torch.as_strided(t0, (rows, D), (D, 1), off)


class _BprBlocksFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n_neg, *blocks):
        emb = adjacent_rows(blocks)
        if emb is None:
            emb = torch.cat(blocks)
        B = blocks[0].shape[0]
        loss, d_emb = bpr_loss_and_grad(emb, B, (len(blocks) - 2) * B, (len(blocks) - 1) * B, n_neg, 1.0)
        ctx.save_for_backward(d_emb)
        ctx.heights = [t.shape[0] for t in blocks]
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d_emb,) = ctx.saved_tensors
        d = d_emb * g
        out, r = [], 0
        for h in ctx.heights:
            out.append(d[r:r + h])
            r += h
        return (None,) + tuple(out)


def bpr_loss_blocks(source_embedding, destination_embedding, negative_embedding, p_pos_embedding=None):
    """The BPR expression of main.py:364-381 (baseline: positive = destination) / 321-337 (``ours``: positive = the p_pos block,
    pass it as ``p_pos_embedding``) on the blocks the reference's loop holds after ``compute_temporal_embeddings[_p]`` - one
    native launch for loss and gradient rows instead of ten torch kernels forward and ~20 backward:

        loss = pfotgnrec_amd.bpr_loss_blocks(source_embedding, destination_embedding, negative_embedding)

    The blocks are used in place when they are the adjacent outputs of one call (no copy either way: the gradient blocks
    go back as adjacent rows of one matrix and the TGN backward takes that matrix as it stands)."""
    B = source_embedding.shape[0]
    if B == 0:
        return source_embedding.sum() * 0.0
    D = source_embedding.shape[-1]
    blocks = [source_embedding.reshape(B, D), destination_embedding.reshape(B, D)]
    if p_pos_embedding is not None:
        blocks.append(p_pos_embedding.reshape(B, D))
    neg = negative_embedding.reshape(-1, D)
    blocks.append(neg)
    return _BprBlocksFn.apply(neg.shape[0] // B, *blocks)


def bpr_step(tgn, emb, batch, n_neg, pos_block=1, grad_scale=None, optimizer=None, collective=None):
    """``loss = bpr_loss(...); loss.backward()`` (main.py:321-337 + 388) as two native calls and no torch kernel: the loss
    kernel writes the already scaled gradient rows and the TGN backward is called on them directly, skipping autograd's
    seed fill and ``d_emb * g`` multiply (three ~6 us launches on the critical path of a 1.5 ms step); the batch mean of the
    per-interaction losses - an input of nothing - is taken beside the backward.  ``emb`` must be the
    tensor ``TGN.embed_device`` returned under autograd; anything else (an empty shard, a view) takes the autograd route.

    ``optimizer`` (a ``FusedAdam`` of this model, single rank): ``loss.backward(); optimizer.step()`` (main.py:388-389) in
    one go, with the END of the backward and the optimizer's kernel left on the library's side stream
    (``pfo_tgn_batch.defer_join`` + ``pfo_tgn_adam_side``): the caller's stream does not wait out the last ~40 us of
    side-stream launches (the chain back to the layer-1 projection weights) nor the Adam kernel - it goes straight on to the
    next batch's candidate draw and neighbour sampling, and the next forward joins where it first needs parameters.
    Gradients and parameters are IN FLIGHT on that stream afterwards: read them only after ``tgn.join()`` (``state_dict()``
    joins by itself).  ``optimizer.zero_grad(set_to_none=True)`` afterwards is fine (host side only).
    ``collective`` (a data-parallel rank, with ``optimizer``): a callable that all-reduces ``tgn.flat_grad`` in place
    (``lambda: allreduce_flat_grad(tgn.flat_grad, world)``).  It is issued on the library's side stream between the end of
    the backward and the optimizer's kernel there: the gradient exchange of step n then runs beside step n+1's candidate
    draw and neighbour sampling instead of holding the caller's stream.  Every rank issues it exactly once per call - also a
    rank whose shard is empty (no native backward: the serial order on the caller's stream, same collective).
    Returns the detached loss."""
    call = getattr(emb.grad_fn, "call", None) if emb.grad_fn is not None else None
    if batch == 0 or call is None or call.ws is None or emb.shape[0] != call.R:
        loss = bpr_loss(emb, batch, n_neg, pos_block, tgn.dp_grad_scale if grad_scale is None else grad_scale)
        loss.backward()
        if optimizer is not None and getattr(optimizer, "tgn", None) is tgn:
            if collective is not None:
                collective()
            optimizer.step()
        return loss.detach()
    scale = tgn.dp_grad_scale if grad_scale is None else grad_scale
    e = emb.detach().contiguous()
    R, D = e.shape
    parts = torch.empty(batch, dtype=torch.float32, device=e.device)
    d_emb = torch.empty_like(e)
    loss = torch.empty(1, dtype=torch.float32, device=e.device)
    _lib.call("pfo_bpr_loss_parts", e.data_ptr(), batch, D, pos_block * batch, (pos_block + 1) * batch, n_neg, R, float(scale),
              parts.data_ptr(), d_emb.data_ptr(), _lib.stream_ptr())
    ours = optimizer is not None and getattr(optimizer, "tgn", None) is tgn
    fused_opt = (ours and (tgn.dp_world == 1 or collective is not None)
                 and not torch.cuda.is_current_stream_capturing() and not _lib.prof_is_on())
    tgn._native_backward(call, d_emb, mean=(parts, loss), defer_join=fused_opt)     # the batch mean of the losses: on the backward's side stream
    call.release()
    if collective is not None and ours:
        if fused_opt:
            with torch.cuda.stream(tgn.side_stream()):      # behind the backward's last side-stream launch, in front of Adam there
                collective()
        else:
            collective()
    if optimizer is not None:
        optimizer.step(side=fused_opt)
    return loss[0]


def time_encode(t, weight, bias):
    """cos(fma(t, w, b)) (model/time_encoding.py:17-25), forward only; t f32[...], returns [..., D]."""
    _lib.require_gpu(t.device)
    t = t.contiguous().float()
    D = bias.shape[0]
    out = torch.empty(t.shape + (D,), dtype=torch.float32, device=t.device)
    _lib.call("pfo_time_encode", t.data_ptr(), t.numel(), weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
              D, out.data_ptr(), _lib.stream_ptr())
    return out


def rank_metrics(emb, batch, n_items):
    """Ranking part of evaluation.py:114-145 on the device: emb [R,D] = [src B | dst B | neg B*n_items].

    Returns (rank i32[B], recall f32[B,3], ndcg f32[B,3]) for k = 1, 3, 5; rank = number of negatives scoring >= the
    positive (canonical tie policy)."""
    _lib.require_gpu(emb.device)
    emb = emb.contiguous()
    D = emb.shape[1]
    rank = torch.empty(batch, dtype=torch.int32, device=emb.device)
    hits = torch.empty((batch, 3), dtype=torch.float32, device=emb.device)
    ndcg = torch.empty((batch, 3), dtype=torch.float32, device=emb.device)
    _lib.call("pfo_rank_metrics", emb.data_ptr(), batch, D, n_items, rank.data_ptr(), hits.data_ptr(), ndcg.data_ptr(),
              _lib.stream_ptr())
    return rank, hits, ndcg
