"""Autograd-facing wrappers of the small native ops (BPR loss, TimeEncode)."""
import torch

from . import _lib


class _BprFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, B, pos_off, neg_off, n_neg, scale):
        _lib.require_gpu(emb.device)
        emb = emb.contiguous()
        R, D = emb.shape
        loss = torch.empty(1, dtype=torch.float32, device=emb.device)
        d_emb = torch.empty_like(emb)
        scratch = torch.empty(B, dtype=torch.float32, device=emb.device)
        _lib.call("pfo_bpr_loss", emb.data_ptr(), B, D, pos_off, neg_off, n_neg, R, float(scale), loss.data_ptr(),
                  d_emb.data_ptr(), scratch.data_ptr(), _lib.stream_ptr())
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
