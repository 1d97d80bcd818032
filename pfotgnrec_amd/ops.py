"""``torch.library`` registration of the stateless native ops (namespace ``pfotgn``).

BASELINE.json's north_star words the boundary as "PyTorch-ROCm custom ops".  The core is the C ABI (``include/pfotgn.h``,
bound in ``_lib.py``); this module registers the entry points that are pure functions of their tensor arguments as
dispatcher ops, so they show up in ``torch.ops.pfotgn.*``, carry fake (meta) implementations for tracing / shape
inference and - for the BPR loss - an autograd formula:

    torch.ops.pfotgn.tnbr_sample(indptr, nbr, eidx, ts, q_nodes, q_ts, K)      utils/utils.py:163-219 (most recent)
    torch.ops.pfotgn.time_encode(t, weight, bias)                              model/time_encoding.py:17-25
    torch.ops.pfotgn.bpr_loss(emb, batch, n_neg, pos_block, grad_scale)        main.py:321-337 / 364-381
    torch.ops.pfotgn.rank_metrics(emb, batch, n_items)                         evaluation.py:114-145

Only a HIP implementation is registered ("cuda" dispatch key = ROCm here): on any other device the dispatcher raises,
there is no CPU fallback.  The TGN step itself keeps its ``autograd.Function`` (``tgn._EmbedFn``): it owns state (memory,
message tables, a workspace per outstanding call) that a functional op schema cannot express.
"""
import torch

from . import _lib, functional

_LIB_NS = "pfotgn"


@torch.library.custom_op(_LIB_NS + "::tnbr_sample", mutates_args=(), device_types="cuda")
def tnbr_sample(indptr: torch.Tensor, adj_nbr: torch.Tensor, adj_eidx: torch.Tensor, adj_ts: torch.Tensor, q_nodes: torch.Tensor,
                q_ts: torch.Tensor, K: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    N = q_nodes.shape[0]
    q_nodes, q_ts = q_nodes.to(torch.int32).contiguous(), q_ts.to(torch.float64).contiguous()
    o_nbr = torch.empty((N, K), dtype=torch.int32, device=q_nodes.device)
    o_eidx = torch.empty((N, K), dtype=torch.int32, device=q_nodes.device)
    o_et = torch.empty((N, K), dtype=torch.float32, device=q_nodes.device)
    _lib.call("pfo_tnbr_sample", _lib.ptr(indptr), _lib.ptr(adj_nbr), _lib.ptr(adj_eidx), _lib.ptr(adj_ts), indptr.shape[0] - 1,
              _lib.ptr(q_nodes), _lib.ptr(q_ts), N, K, 0, None, 0, 0, _lib.ptr(o_nbr), _lib.ptr(o_eidx), _lib.ptr(o_et), None,
              None, None, _lib.stream_ptr())
    return o_nbr, o_eidx, o_et


@tnbr_sample.register_fake
def _(indptr, adj_nbr, adj_eidx, adj_ts, q_nodes, q_ts, K):
    N = q_nodes.shape[0]
    return (q_nodes.new_empty((N, K), dtype=torch.int32), q_nodes.new_empty((N, K), dtype=torch.int32),
            q_nodes.new_empty((N, K), dtype=torch.float32))


@torch.library.custom_op(_LIB_NS + "::time_encode", mutates_args=(), device_types="cuda")
def time_encode(t: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    return functional.time_encode(t, weight.reshape(-1), bias)


@time_encode.register_fake
def _(t, weight, bias):
    return t.new_empty(t.shape + (bias.shape[0],), dtype=torch.float32)


@torch.library.custom_op(_LIB_NS + "::bpr_loss_fwd", mutates_args=(), device_types="cuda")
def _bpr_loss_fwd(emb: torch.Tensor, batch: int, n_neg: int, pos_block: int, grad_scale: float) -> tuple[torch.Tensor, torch.Tensor]:
    emb = emb.contiguous()
    R, D = emb.shape
    loss = torch.empty(1, dtype=torch.float32, device=emb.device)
    d_emb = torch.empty_like(emb)
    scratch = torch.empty(batch, dtype=torch.float32, device=emb.device)
    _lib.call("pfo_bpr_loss", emb.data_ptr(), batch, D, pos_block * batch, (pos_block + 1) * batch, n_neg, R, float(grad_scale),
              loss.data_ptr(), d_emb.data_ptr(), scratch.data_ptr(), _lib.stream_ptr())
    return loss.reshape(()), d_emb


@_bpr_loss_fwd.register_fake
def _(emb, batch, n_neg, pos_block, grad_scale):
    return emb.new_empty(()), torch.empty_like(emb)


def _bpr_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _bpr_backward(ctx, g_loss, g_demb):
    (d_emb,) = ctx.saved_tensors
    return d_emb * g_loss, None, None, None, None


_bpr_loss_fwd.register_autograd(_bpr_backward, setup_context=_bpr_setup)


def bpr_loss(emb, batch, n_neg, pos_block=1, grad_scale=1.0):
    """Dispatcher form of ``pfotgnrec_amd.bpr_loss`` (same semantics)."""
    return torch.ops.pfotgn.bpr_loss_fwd(emb, batch, n_neg, pos_block, grad_scale)[0]


@torch.library.custom_op(_LIB_NS + "::rank_metrics", mutates_args=(), device_types="cuda")
def rank_metrics(emb: torch.Tensor, batch: int, n_items: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return functional.rank_metrics(emb, batch, n_items)


@rank_metrics.register_fake
def _(emb, batch, n_items):
    return (emb.new_empty((batch,), dtype=torch.int32), emb.new_empty((batch, 3), dtype=torch.float32),
            emb.new_empty((batch, 3), dtype=torch.float32))
