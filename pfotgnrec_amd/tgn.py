"""TGN facade - drop-in for the reference's ``model/tgn.py`` on the training path.

Same constructor and the same two entry points ``main.py`` / ``evaluation.py`` use
(``compute_temporal_embeddings`` tgn.py:219, ``compute_temporal_embeddings_p`` tgn.py:102), same
parameter names in ``state_dict()``; underneath, one flat fp32 parameter buffer in HBM and three
native calls per step (``pfo_tgn_forward`` / ``pfo_tgn_update_state`` / ``pfo_tgn_backward``).
PyTorch supplies device memory, the autograd hand-off and the optimizer interface - no torch op
runs on the hot path.  Without a HIP device every compute method raises; there is no CPU path.

Supported (what the training path of main.py:106-121 selects): ``embedding_module_type =
"graph_attention"``, ``message_function = "identity"``, ``aggregator_type = "last"``,
``memory_updater_type = "gru"``, ``use_memory`` True (TGN / ours) or False (TGAT, main.py:70-74).
Other enum values raise ``ValueError`` like the reference's factories do for unknown names.
"""
import ctypes
import os
import math

import numpy as np
import torch
from torch import nn

from . import _lib
from .memory import Memory


class _Holder(nn.Module):
    """Bare container used to reproduce the reference's parameter names."""


def _normalise_edge_features(edge_features):
    # tgn.py:38-41: fp32 z-score per column, the padding row 0 included
    ef = np.asarray(edge_features).astype(np.float32)
    ef -= ef.mean(axis=0)
    ef /= ef.std(axis=0)
    return ef.astype(np.float32)


class _Call:
    """Everything one forward/backward pair of native calls needs to agree on."""
    __slots__ = ("roots", "root_ts", "R", "K", "mode", "draws", "draw_ptrs", "keep_ptrs", "seed", "offset", "dropout_p", "training",
                 "extra", "batch_struct", "ws", "ws_caps", "cfg", "pool", "gru_applied", "ready", "keep", "pkey", "__weakref__")

    def release(self):
        """Hands the call's workspace back to its TGN's pool (after the backward, or when the graph is dropped)."""
        ws, pool = getattr(self, "ws", None), getattr(self, "pool", None)
        if ws is not None and pool is not None:
            pool.append((self.ws_caps, ws))
        self.ws = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


_TORCH_OF = {"float64": torch.float64, "float32": torch.float32, "int32": torch.int32, "int64": torch.int64, "uint8": torch.uint8}


class _EmbedFn(torch.autograd.Function):
    """Autograd bridge: forward = native forward (+ state update), backward = native backward.

    Parameter gradients are accumulated by the native call straight into the flat gradient buffer
    that every ``p.grad`` is a view of, so ``None`` is returned for the parameter inputs.
    """

    @staticmethod
    def forward(ctx, tgn, call, post, split, *params):
        # ``split`` = row ranges ((r0, r1), ...) of the embedding matrix handed out as SEPARATE outputs (the reference's entry
        # points return (source, destination, negative ...) embeddings, tgn.py:215-217 / 325-327): as outputs of this node their
        # gradients arrive side by side and one concatenation rebuilds d emb - as three slices of one output autograd filled
        # three zero matrices of the full size, copied a block into each and added them up (8 launches, ~30 us of device time
        # and ~60 us of host time per batch on the drop-in loop)
        emb = tgn._native_forward(call)
        if post is not None:
            post(call)
        ctx.tgn, ctx.call, ctx.n_params, ctx.split = tgn, call, len(params), split
        if split is None:
            return emb
        return tuple(emb[r0:r1] for r0, r1 in split)

    @staticmethod
    def backward(ctx, *d_out):
        if ctx.call.ws is None:
            raise RuntimeError("backward through a TGN forward whose workspace was already consumed "
                               "(a second backward over the same call is not supported)")
        if ctx.split is None:
            d_emb = d_out[0].contiguous()
        else:
            from .functional import adjacent_rows
            d_emb = adjacent_rows(d_out)              # (bpr_loss_blocks hands back the cut of one matrix: used as it stands)
            if d_emb is None:
                d_emb = torch.cat(d_out)
        if ctx.tgn._overlap_ok():
            ctx.tgn._backward_beside(ctx.call, d_emb)
        else:
            ctx.tgn._native_backward(ctx.call, d_emb)
        ctx.call.release()
        return (None, None, None, None) + (None,) * ctx.n_params


class TGN(nn.Module):
    def __init__(self, neighbor_finder, node_features, edge_features, device, n_layers=2, n_heads=2, dropout=0.1,
                 use_memory=False, memory_update_at_start=True, message_dimension=100, memory_dimension=500,
                 embedding_module_type="graph_attention", message_function="mlp", mean_time_shift_src=0,
                 std_time_shift_src=1, mean_time_shift_dst=0, std_time_shift_dst=1, n_neighbors=None,
                 aggregator_type="last", memory_updater_type="gru", use_destination_embedding_in_message=False,
                 use_source_embedding_in_message=False, dyrep=False):
        super().__init__()
        if embedding_module_type != "graph_attention":
            raise ValueError("Embedding Module {} not supported".format(embedding_module_type))
        if use_memory:
            if message_function != "identity":
                raise ValueError("Message function {} not supported (identity only)".format(message_function))
            if aggregator_type != "last":
                raise ValueError("Message aggregator {} not implemented".format(aggregator_type))
            if memory_updater_type != "gru":
                raise ValueError("Memory updater {} not supported (gru only)".format(memory_updater_type))
            if not memory_update_at_start:
                raise ValueError("memory_update_at_start=False is not supported")
        if dyrep or use_destination_embedding_in_message or use_source_embedding_in_message:
            raise ValueError("dyrep / embedding-in-message variants are not supported")

        self.device = torch.device(device)
        self.n_layers, self.n_heads, self.dropout = int(n_layers), int(n_heads), float(dropout)
        self.use_memory = bool(use_memory)
        self.memory_update_at_start = True
        self.n_neighbors = n_neighbors
        self.embedding_module_type = embedding_module_type
        self.dyrep = False
        self.mean_time_shift_src, self.std_time_shift_src = mean_time_shift_src, std_time_shift_src
        self.mean_time_shift_dst, self.std_time_shift_dst = mean_time_shift_dst, std_time_shift_dst

        self.node_raw_features = torch.from_numpy(np.asarray(node_features).astype(np.float32)).to(self.device).contiguous()
        self.edge_raw_features = torch.from_numpy(_normalise_edge_features(edge_features)).to(self.device).contiguous()
        self.n_nodes, self.n_node_features = self.node_raw_features.shape
        self.n_edge_features = self.edge_raw_features.shape[1]
        self.embedding_dimension = self.n_node_features
        D, Ef = self.n_node_features, self.n_edge_features
        if self.use_memory and memory_dimension != D:
            raise ValueError("memory_dimension must equal the node-feature dimension (embedding_module.py:98)")
        self.memory_dimension = D

        self._cfg = _lib.TgnConfig(self.n_nodes, self.edge_raw_features.shape[0], D, Ef, self.n_layers, self.n_heads,
                                   int(self.use_memory), 1, 1, 1)
        self._layout = _lib.TgnLayout()
        _lib.call("pfo_tgn_param_layout", ctypes.byref(self._cfg), ctypes.byref(self._layout))
        self._flat = torch.zeros(self._layout.total, dtype=torch.float32, device=self.device)
        self._flat_grad = None
        self._views = []          # (parameter, offset, numel, shape)
        self._adj_cache = None
        self.dp_grad_scale = 1.0
        self._ws_pool = []        # free workspaces: [(caps, tensor)]
        self._side_stream = None
        self._prefetched, self._pre_stream, self._pre_main = None, None, None
        self.fuse_state_update = os.environ.get("PFO_FUSE_STATE", "1") != "0"      # training calls: persist + message store inside the native forward, on its side stream
        self._ws_caps = (0, 0, 0)
        # Parameter cache (pfo_tgn_state.pcache): composite weights + fp16 weight images, rebuilt only when the parameters
        # changed.  "Changed" = the torch version counters of the flat buffer AND of every nn.Parameter (in-place ops of any
        # torch optimizer, load_state_dict, copy_ ... bump them; after a real device move the parameters keep counters of their
        # own - ``p.data = view`` does not share the new buffer's) or ``parameters_changed()`` (native writers: FusedAdam,
        # graph replays).  Writes through ``p.data`` bypass the counters: call ``parameters_changed()`` after them.
        self.param_cache = os.environ.get("PFO_PCACHE", "1") != "0"
        # FusedAdam may rebuild the cache right behind its kernel (pfo_tgn_refresh, second side stream) instead of leaving it to
        # the next forward.  Off by default: measured at C2, 1.461 ms per step with, 1.454 without - the ~15 launches are hidden
        # beside the sampling / GRU phase either way, and the refresh adds two event waits and moves the GRU's image launch
        self.refresh_after_step = os.environ.get("PFO_PCACHE_REFRESH", "0") != "0"
        self._pcache, self._pcache_key, self._param_epoch = None, None, 0
        self._last_ws = None      # (config, workspace) of the newest forward (debug_touched)
        self._step = 0
        self.seed = 0
        self.dp_rank, self.dp_world = 0, 1
        self.deterministic = False        # bitwise run-to-run reproducible backward (pfo_tgn_batch.deterministic), ~4 % slower
        self.dp_bucketed = False          # ask the backward for the "top layer's gradients are final" event (two-bucket all-reduce)
        self.dp_ordered = False           # fused step: reduce + optimizer step per bucket, first-use bucket first (distributed.allreduce_flat_grad_ordered)
        self._comm_pending = None         # the communication stream the top block's all-reduce of this step is in flight on
        self._bucket_event, self._bucket_event_fresh, self._grad_split = None, False, None
        self._mid_event, self._mid_event_fresh = None, False   # recorded by the native backward in front of layer 1's attention backward
        self.seg_in_forward_dp = os.environ.get("PFO_SEG_FWD_DP", "1") != "0"   # ... also on a data-parallel rank (A/B switch; emulated rank 0 of 8: 1.4465 -> 1.439 ms)
        self.seg_in_forward = os.environ.get("PFO_SEG_FWD", "1") != "0"      # the backward's instance groups are built beside the forward's layer 1
        self.record_mid_event = False     # ... only on request: an event record on the caller's stream costs the step a launch gap
        self.mid_event_late = False       # record it behind the attention backward instead of in front of it
        self._zero_next = False
        self._grad_zeroed = False         # the optimizer's kernel cleared the flat gradient buffer (FusedAdam(zero_grads_in_step=True))
        # FusedAdam(tgn, overlap_backward=True): loss.backward() leaves the native backward - and optimizer.step() its kernel - on
        # a stream of their own, so that the loop's per-batch ``loss.item()`` (main.py:390) waits for the forward only
        self.overlap_backward = False
        self._bwd_stream, self._bwd_event, self._bwd_hold, self._bwd_joined, self._bwd_home = None, None, [], set(), None
        self.eval_chunk_roots = 16384     # roots per forward-only pass (evaluation.py scores B*(2+N_ITEMS) roots per batch)
        self.eval_dedup = True            # forward-only passes embed every distinct (node, time) root once
        # memory_updater.py:25,41 assert that no pending message is older than its node's last update; the check reads
        # device state back (one sync per call), so it is off unless asked for
        self.debug_checks = False

        E, C, M = 2 * D, 2 * D + Ef, 3 * D + Ef
        lay = self._layout
        self.time_encoder = _Holder()
        self.time_encoder.dimension = D
        self.time_encoder.w = _Holder()
        self._register(self.time_encoder.w, "weight", lay.time_w, (D, 1))
        self._register(self.time_encoder.w, "bias", lay.time_b, (D,))

        self.memory = None
        self._gru_params = set()
        self._gru_applied_now = True
        if self.use_memory:
            self.memory = Memory(n_nodes=self.n_nodes, memory_dimension=D, input_dimension=M, message_dimension=M,
                                 device=self.device)
            self.memory_updater = _Holder()
            self.memory_updater.layer_norm = nn.LayerNorm(D).to(self.device)     # constructed, never applied (memory_updater.py:14)
            self.memory_updater.memory_updater = _Holder()
            gru = self.memory_updater.memory_updater
            self._register(gru, "weight_ih", lay.gru_w_ih, (3 * D, M))
            self._register(gru, "weight_hh", lay.gru_w_hh, (3 * D, D))
            self._register(gru, "bias_ih", lay.gru_b_ih, (3 * D,))
            self._register(gru, "bias_hh", lay.gru_b_hh, (3 * D,))
            self._gru_params = set(gru.parameters())

        self.embedding_module = _Holder()
        self.embedding_module.neighbor_finder = neighbor_finder       # main.py:427 assigns this attribute directly
        self.embedding_module.attention_models = nn.ModuleList()
        for l in range(self.n_layers):
            q = lay.layer[l]
            att = _Holder()
            att.multi_head_target = _Holder()
            mha = att.multi_head_target
            self._register(mha, "q_proj_weight", q.wq, (E, E))
            self._register(mha, "k_proj_weight", q.wk, (E, C))
            self._register(mha, "v_proj_weight", q.wv, (E, C))
            self._register(mha, "in_proj_bias", q.b_in, (3 * E,))
            mha.out_proj = _Holder()
            self._register(mha.out_proj, "weight", q.wo, (E, E))
            self._register(mha.out_proj, "bias", q.bo, (E,))
            att.merger = _Holder()
            att.merger.fc1, att.merger.fc2 = _Holder(), _Holder()
            self._register(att.merger.fc1, "weight", q.w1, (D, E + D))
            self._register(att.merger.fc1, "bias", q.b1, (D,))
            self._register(att.merger.fc2, "weight", q.w2, (D, D))
            self._register(att.merger.fc2, "bias", q.b2, (D,))
            self.embedding_module.attention_models.append(att)
        self.reset_parameters()

    # ------------------------------------------------------------------ parameters
    def _register(self, module, name, offset, shape):
        n = int(np.prod(shape))
        p = nn.Parameter(self._flat[offset:offset + n].view(shape))
        module.register_parameter(name, p)
        self._views.append((p, int(offset), n, tuple(shape)))

    def reset_parameters(self):
        """The reference's initialisers: TimeEncode (time_encoding.py:13-15), nn.GRUCell, nn.MultiheadAttention
        (xavier-uniform in-projections, zero biases), MergeLayer xavier-normal (utils.py:11-12)."""
        D = self.n_node_features
        with torch.no_grad():
            te = self.time_encoder.w
            te.weight.copy_(torch.from_numpy((1 / 10 ** np.linspace(0, 9, D)).astype(np.float32)).reshape(D, 1))
            te.bias.zero_()
            if self.use_memory:
                k = 1.0 / math.sqrt(D)
                for p in self.memory_updater.memory_updater.parameters():
                    p.uniform_(-k, k)
            for att in self.embedding_module.attention_models:
                mha = att.multi_head_target
                for w in (mha.q_proj_weight, mha.k_proj_weight, mha.v_proj_weight):
                    nn.init.xavier_uniform_(w)
                mha.in_proj_bias.zero_()
                nn.init.kaiming_uniform_(mha.out_proj.weight, a=math.sqrt(5))
                mha.out_proj.bias.zero_()
                for fc in (att.merger.fc1, att.merger.fc2):
                    nn.init.xavier_normal_(fc.weight)
                    bound = 1.0 / math.sqrt(fc.weight.shape[1])
                    fc.bias.uniform_(-bound, bound)

    def hot_parameters(self):
        return [v[0] for v in self._views]

    @property
    def flat_parameters(self):
        return self._flat

    @property
    def flat_grad(self):
        return self._flat_grad

    def _apply(self, fn, recurse=True):
        self.join()                                               # a side-stream optimizer step may still be writing the buffer
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("the native path is fp32 only (1e-4 parity bar)")
        self._flat = new_flat.contiguous()
        self._flat_grad = None
        self._side_stream = None
        for p, off, n, shape in self._views:
            p.data = self._flat[off:off + n].view(shape)
            p.grad = None
        self.node_raw_features = fn(self.node_raw_features).contiguous()
        self.edge_raw_features = fn(self.edge_raw_features).contiguous()
        if self.use_memory:
            mem = self.memory
            with torch.no_grad():
                mem.memory.data = fn(mem.memory.data)
                mem.last_update.data = fn(mem.last_update.data)
            mem.msg_table, mem.msg_time, mem.has_msg = fn(mem.msg_table), fn(mem.msg_time), mem.has_msg.to(self._flat.device)
            mem.device = self._flat.device
            self.memory_updater.layer_norm._apply(fn)
        self.device = self._flat.device
        if self._prefetched is not None and self._prefetched[2].ws is not None and self._prefetched[2].ws.is_cuda:
            torch.cuda.current_stream(self._prefetched[2].ws.device).wait_event(self._prefetched[2].ready)
        self._ws_pool, self._last_ws, self._adj_cache = [], None, None
        self._prefetched = None
        self._pcache, self._pcache_key = None, None
        return self

    # ------------------------------------------------------------------ neighbour finder plumbing
    @property
    def neighbor_finder(self):
        return self.embedding_module.neighbor_finder

    @neighbor_finder.setter
    def neighbor_finder(self, nf):
        self.embedding_module.neighbor_finder = nf

    def set_neighbor_finder(self, neighbor_finder):
        """tgn.py:380-382."""
        self.embedding_module.neighbor_finder = neighbor_finder

    def set_data_parallel(self, rank, world_size):
        """Edge-batch data parallelism: this rank embeds interactions [rank*B/W, (rank+1)*B/W)."""
        self.dp_rank, self.dp_world = int(rank), int(world_size)

    # ------------------------------------------------------------------ native plumbing
    def _acquire_workspace(self, R, K, B):
        """A workspace (and the config it was carved for) for ONE forward[/backward] pair.

        Every outstanding training forward owns its workspace until its backward has run, so several forwards may
        precede one backward (main.py:171 accumulates ``BACKPROP_EVERY`` batches) and an evaluation forward between a
        training forward and its backward cannot clobber saved activations.  Workspaces are pooled: in the usual
        one-forward-one-backward loop the same buffer (same device pointers) is reused every step.
        """
        need = (max(1, int(R)), max(1, int(K)), max(1, int(B)))
        best = None
        for i, (caps, _) in enumerate(self._ws_pool):
            if all(c >= n for c, n in zip(caps, need)) and (best is None or caps < self._ws_pool[best][0]):
                best = i
        if best is not None:
            caps, ws = self._ws_pool.pop(best)
        else:
            caps = tuple(max(c, n) for c, n in zip(self._ws_caps, need))
            self._ws_caps = caps
            self._ws_pool[:] = [e for e in self._ws_pool if e[0] == caps]   # smaller buffers are not worth keeping (in place:
                                                                            # outstanding calls hand theirs back to this list)
            cfg = self._cfg_for(caps)
            nbytes = _lib.load().pfo_tgn_workspace_bytes(ctypes.byref(cfg))
            if nbytes < 0:
                raise _lib.PfoError("pfo_tgn_workspace_bytes: %s" % _lib.load().pfo_last_error().decode())
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return caps, self._cfg_for(caps), ws

    def _cfg_for(self, caps):
        c = self._cfg
        return _lib.TgnConfig(c.n_nodes, c.n_edges_p1, c.D, c.Ef, c.n_layers, c.n_heads, c.use_memory, caps[0], caps[1], caps[2])

    def _param_key(self):
        # every parameter's own counter too: after .to(device) / .float() they no longer share the flat buffer's
        return (self._flat._version, sum(v[0]._version for v in self._views), self._param_epoch, self._flat.data_ptr())

    def parameters_changed(self, refresh=False):
        """Tells the model that something outside torch's view wrote the parameters (a native optimizer kernel, a graph replay,
        a write through ``p.data``): the parameter cache is rebuilt by the next forward - or, with ``refresh``, right now on
        the library's side stream (``pfo_tgn_refresh``), so that the next forward launches none of it."""
        self._param_epoch += 1
        self._pcache_key = None
        if (refresh and self.param_cache and self._pcache is not None and self._flat.is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            st = self._state_struct(adjacency=False)
            _lib.call("pfo_tgn_refresh", ctypes.byref(self._cfg), ctypes.byref(st), _lib.stream_ptr())
            self._pcache_key = self._param_key()

    def _state_struct(self, adjacency=True):
        if adjacency:
            indptr, nbr, eidx, ts = self._adjacency()
            self._keepalive = (indptr, nbr, eidx, ts)
            adj = (indptr.data_ptr(), nbr.data_ptr(), eidx.data_ptr(), ts.data_ptr())
        else:
            adj = (None, None, None, None)
        mem = self.memory
        pc, valid = None, 0
        if self.param_cache and self._flat.is_cuda:
            if self._pcache is None:
                nbytes = _lib.load().pfo_tgn_pcache_bytes(ctypes.byref(self._cfg))
                if nbytes < 0:
                    raise _lib.PfoError("pfo_tgn_pcache_bytes: %s" % _lib.load().pfo_last_error().decode())
                self._pcache, self._pcache_key = torch.zeros(nbytes, dtype=torch.uint8, device=self.device), None   # (zeroed ONCE: include/pfotgn.h)
            pc = self._pcache.data_ptr()
            # (a step being captured into a HIP graph always builds: its replays run with whatever the parameters are then)
            valid = int(self._pcache_key is not None and self._pcache_key == self._param_key()
                        and not torch.cuda.is_current_stream_capturing())
        return _lib.TgnState(*adj, self.node_raw_features.data_ptr(), self.edge_raw_features.data_ptr(),
                             mem.memory.data_ptr() if mem is not None else None,
                             mem.last_update.data_ptr() if mem is not None else None,
                             mem.msg_table.data_ptr() if mem is not None else None,
                             mem.msg_time.data_ptr() if mem is not None else None,
                             mem.has_msg.data_ptr() if mem is not None else None, self._flat.data_ptr(), pc, valid)

    def _adjacency(self):
        """Device CSR of the current neighbour finder, checked against this model's node table.

        The native step indexes ``indptr`` with every node id below ``n_nodes`` (= rows of the node-feature table) and
        the feature / memory tables with every neighbour id the CSR returns.  ``get_neighbor_finder(data, uniform)``
        (main.py:95) sizes its adjacency by the largest id *in that split*, which can be smaller: such a finder gets
        its ``indptr`` padded with empty rows (the reference would raise IndexError for those ids).  A finder that
        names nodes this model has no features for is refused.
        """
        nf = self.neighbor_finder
        key = (id(nf), str(self.device), getattr(nf, "_version", 0))        # append() bumps the version: new arrays
        if self._adj_cache is not None and self._adj_cache[0] == key and self._adj_cache[1] is nf:
            return self._adj_cache[2]
        indptr, nbr, eidx, ts = nf.device_arrays(self.device)
        max_nbr = nf.max_neighbor_id() if hasattr(nf, "max_neighbor_id") else int(np.max(nf.nbr, initial=0))
        extra_rows_used = False
        if nf.n_nodes > self.n_nodes:                    # rows beyond this model's node table must be empty
            end_mine, end_all = nf.rows_end(self.n_nodes) if hasattr(nf, "rows_end") else (int(nf.indptr[self.n_nodes]), int(nf.indptr[-1]))
            extra_rows_used = end_mine != end_all
        if max_nbr >= self.n_nodes or extra_rows_used:
            raise ValueError("neighbour finder references node ids >= n_nodes (%d): node features have %d rows"
                             % (max(max_nbr, nf.n_nodes - 1), self.n_nodes))
        max_eidx = nf.max_edge_idx() if hasattr(nf, "max_edge_idx") else int(np.max(nf.eidx, initial=0))
        if max_eidx >= self.edge_raw_features.shape[0]:
            raise ValueError("neighbour finder references edge index %d but edge features have %d rows"
                             % (max_eidx, self.edge_raw_features.shape[0]))
        if indptr.shape[0] < self.n_nodes + 1:
            pad = indptr[-1:].expand(self.n_nodes + 1 - indptr.shape[0])
            indptr = torch.cat([indptr, pad]).contiguous()
        self._adj_cache = (key, nf, (indptr, nbr, eidx, ts))
        return self._adj_cache[2]

    def _make_call(self, roots, root_ts, K, draws, dropout_p, extra, B, offset_dev=None, defer_step=False, dropout_keep=None):
        c = _Call()
        c.roots, c.root_ts, c.R, c.K = roots, root_ts, int(roots.shape[0]), int(K)
        uniform = bool(getattr(self.neighbor_finder, "uniform", False))
        c.mode = 0 if not uniform else (1 if draws is not None else 2)
        c.draws = draws
        c.draw_ptrs = None
        if draws is not None:
            if len(draws) != self.n_layers:
                raise ValueError("uniform mode with injected draws needs one index tensor per layer")
            c.draw_ptrs = (ctypes.c_void_p * self.n_layers)(*[d.data_ptr() for d in draws])
        if not defer_step:                # (a call prepared ahead of time takes its place in the sequence when it is used)
            self._step += 1
        c.seed = self.seed + getattr(self.neighbor_finder, "seed", 0)
        # position in the Philox streams (dropout masks, uniform draws): the call counter - or, for a step captured into a HIP
        # graph (whose kernel arguments are frozen), a device word the graph itself advances
        c.offset = (self._step << 36) if (offset_dev is None and not defer_step) else 0
        c.dropout_p = float(dropout_p)
        c.training = int(dropout_p > 0.0)
        c.extra = extra
        c.batch_struct = _lib.TgnBatch(roots.data_ptr(), root_ts.data_ptr(), c.R, c.K, c.mode,
                                       ctypes.cast(c.draw_ptrs, ctypes.POINTER(ctypes.c_void_p)) if c.draw_ptrs else None,
                                       c.seed, c.offset, c.dropout_p, c.training,
                                       extra.data_ptr() if extra is not None else None,
                                       int(extra.shape[0]) if extra is not None else 0,
                                       offset_dev.data_ptr() if offset_dev is not None else None,
                                       1 if self.deterministic else 0)
        c.keep_ptrs = None
        if dropout_keep is not None:
            # injected dropout decisions (parity tests): one u8 [n_l, K] tensor per layer, the roots' level first (like draws)
            if len(dropout_keep) != self.n_layers:
                raise ValueError("injected dropout decisions need one uint8 tensor per layer")
            n = c.R
            for t in dropout_keep:
                if t.dtype != torch.uint8 or tuple(t.shape) != (n, c.K) or not t.is_contiguous():
                    raise ValueError("dropout_keep tensors must be contiguous uint8 [n_l, K] in level order (roots first)")
                n *= 1 + c.K
            c.keep = (getattr(c, "keep", None), dropout_keep)
            c.keep_ptrs = (ctypes.c_void_p * self.n_layers)(*[t.data_ptr() for t in dropout_keep])
            c.batch_struct.dropout_keep = ctypes.cast(c.keep_ptrs, ctypes.POINTER(ctypes.c_void_p))
        c.pool = self._ws_pool
        c.ws_caps, c.cfg, c.ws = self._acquire_workspace(c.R, c.K, B)
        c.gru_applied = self._gru_applied_now
        return c

    def _torch_versions(self):
        return (self._flat._version, sum(v[0]._version for v in self._views))

    def _native_forward(self, call, out=None):
        _lib.require_gpu(self.device)
        self._join_backward()
        # torch wrote the parameters on the caller's stream since the last native forward (p.copy_(), a torch optimizer ...):
        # that write is ordered against nothing the library left on its side stream, and the forward would not fork from
        # the caller's stream while a deferred step is pending there - join first (a no-op when nothing is pending)
        tv = self._torch_versions()
        if tv != getattr(self, "_seen_versions", None):
            self._seen_versions = tv
            if not torch.cuda.is_current_stream_capturing():
                self.join()
        st = self._state_struct()
        emb = out if out is not None else torch.empty((call.R, self.n_node_features), dtype=torch.float32, device=self.device)
        _lib.call("pfo_tgn_forward", ctypes.byref(call.cfg), ctypes.byref(st), ctypes.byref(call.batch_struct),
                  call.ws.data_ptr(), emb.data_ptr(), _lib.stream_ptr())
        if st.pcache:
            if torch.cuda.is_current_stream_capturing():
                self._pcache_key = None                            # (nothing ran: the capture's replays will build it)
            elif not st.pcache_valid:
                self._pcache_key = self._param_key()               # this call built the cache for the current parameters
            call.pkey = self._param_key()
        else:
            call.pkey = None
        self._last_ws = (call.cfg, call.ws)
        self._last_call = (int(call.seed), int(call.batch_struct.offset), int(call.R), int(call.K), float(call.dropout_p),
                           call.batch_struct.offset_dev is not None)
        return emb

    def _attach_grads(self, gru_applied=True, defer_zero=False):
        """Every ``p.grad`` becomes a view of the flat gradient buffer the native backward accumulates into.  When the
        GRU was not applied in the forward (no node held a pending message: the first batch after ``__init_memory__``)
        the reference's autograd leaves the four GRU tensors' ``.grad`` at None (memory_updater.py:38-40 returns before
        the cell is called) and torch.optim.Adam skips them; the same is done here.

        Returns True when the WHOLE buffer still has to be cleared and ``defer_zero`` asked to leave that to the native
        backward (which does it on its side stream, off the critical path)."""
        if self._flat_grad is None:
            self._flat_grad = torch.zeros_like(self._flat)
        gv = self.__dict__.get("_grad_views")
        if gv is None or gv[0] is not self._flat_grad:
            # the views are made once per buffer: main.py's optimizer.zero_grad() drops every .grad in front of every batch, and
            # slicing + reshaping ~50 views again was 70 us of host time per batch in front of the native backward
            gv = self.__dict__["_grad_views"] = (self._flat_grad, [self._flat_grad[off:off + n].view(shape) for _, off, n, shape in self._views])
        missing = [i for i, v in enumerate(self._views) if v[0].grad is None]
        deferred = False
        if len(missing) == len(self._views):
            if defer_zero:
                deferred = True
            else:
                self._flat_grad.zero_()
        else:
            for i in missing:
                gv[1][i].zero_()
        for i in missing:
            p = self._views[i][0]
            if not gru_applied and p in self._gru_params:
                continue
            p.grad = gv[1][i]
        return deferred

    def side_stream(self):
        """The library's first side stream as a torch stream (``pfo_tgn_side_stream``): work a caller must order between a
        deferred backward end and the optimizer step there - a data-parallel rank's gradient all-reduce (``bpr_step``)."""
        if self._side_stream is None:
            ptr = _lib.load().pfo_tgn_side_stream()
            if not ptr:
                raise _lib.PfoError("pfo_tgn_side_stream: %s" % _lib.load().pfo_last_error().decode())
            self._side_stream = torch.cuda.ExternalStream(ptr, device=self.device)
        return self._side_stream

    def wait_comm_stream(self):
        """The library's side stream waits for the top block's all-reduce (``allreduce_flat_grad_ordered``), once per step."""
        comm, self._comm_pending = self._comm_pending, None
        if comm is not None:
            self.side_stream().wait_stream(comm)

    def join(self):
        """Makes the current stream wait for a backward end / optimizer step that ``bpr_step(..., optimizer=...)`` left on the
        library's side stream, and for a whole backward + step that ``overlap_backward`` left on theirs (no-op when nothing is
        pending)."""
        if self._flat.is_cuda:
            self._join_backward()
            _lib.call("pfo_tgn_join", _lib.stream_ptr())

    # ------------------------------------------------------------------ backward beside the host loop (overlap_backward)
    def _overlap_ok(self):
        return (self.overlap_backward and self.dp_world == 1 and self._flat.is_cuda and not self.record_mid_event
                and not self.dp_bucketed and not torch.cuda.is_current_stream_capturing())

    def _backward_stream(self):
        if self._bwd_stream is None:
            self._bwd_stream = torch.cuda.Stream(device=self.device)
        return self._bwd_stream

    def _join_backward(self):
        """The current stream waits for what ``overlap_backward`` left in flight (once per stream and event).  The tensors that
        work reads (gradient rows, the call's roots) are let go when the stream they were allocated on has waited: the
        allocator may then hand their memory to that stream again."""
        ev = self._bwd_event
        if ev is None:
            return
        key = _lib.stream_ptr()
        if key in self._bwd_joined or (self._bwd_stream is not None and key == self._bwd_stream.cuda_stream):
            return                                                # (the backward stream itself: the optimizer's kernel behind the backward)
        _lib.current_stream().wait_event(ev)
        self._bwd_joined.add(key)
        if key == self._bwd_home:
            self._bwd_hold = []

    def _set_backward_event(self, ev):
        self._bwd_event, self._bwd_joined = ev, set()

    def _backward_beside(self, call, d_emb):
        """``_native_backward`` on the backward stream, behind everything the current stream holds (the gradient rows)."""
        cur, side = _lib.current_stream(), self._backward_stream()
        fork = torch.cuda.Event()
        fork.record(cur)
        side.wait_event(fork)
        with _lib.on_stream(side):
            self._native_backward(call, d_emb)
            ev = torch.cuda.Event()
            ev.record(side)
        self._bwd_hold.append((d_emb, call.roots, call.root_ts, call.extra, getattr(call, "keep", None), call.draws))
        self._bwd_home = cur.cuda_stream
        self._set_backward_event(ev)

    def state_dict(self, *args, **kwargs):
        self.join()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        # the copy runs on the caller's stream: it must not race a side-stream optimizer step, and the next forward must not
        # skip its fork (the parameters were written on THIS stream)
        self.join()
        out = super().load_state_dict(*args, **kwargs)
        self.parameters_changed()
        return out

    def _native_backward(self, call, d_emb, mean=None, defer_join=False):
        """``mean`` = (src f32[n], out f32[1]): a mean the backward takes on its side stream (the BPR loss value)."""
        self._join_backward()             # (a no-op on the backward stream itself and when nothing is in flight)
        zero_first = self._attach_grads(call.gru_applied, defer_zero=True) or self._zero_next
        self._zero_next = False
        if self._grad_zeroed:             # (the buffer IS clear: the optimizer's side-stream kernel wrote the zeros behind its reads)
            zero_first = False
        self._grad_zeroed = False         # this backward accumulates into it
        st = self._state_struct()
        pkey = getattr(call, "pkey", None)
        if pkey is not None and not torch.cuda.is_current_stream_capturing() and pkey != self._param_key():
            # the composites the backward reads (parameter cache) belong to the parameters of the forward
            raise RuntimeError("the parameters were modified between a TGN forward and its backward "
                               "(optimizer step / load_state_dict in between): run the backward first")
        ev = None
        if self.dp_bucketed and self.grad_split < self._layout.total:
            # the top layer's gradient block is final when this event fires (distributed.allreduce_flat_grad_buckets)
            if self._bucket_event is None:
                self._bucket_event = torch.cuda.Event()
                self._bucket_event.record()                       # materialises the underlying hipEvent_t
            ev = self._bucket_event.cuda_event
            self._bucket_event_fresh = True
        if self.record_mid_event and not torch.cuda.is_current_stream_capturing():
            if self._mid_event is None:
                self._mid_event = torch.cuda.Event()
                self._mid_event.record()                          # materialises the underlying hipEvent_t
            call.batch_struct.mid_event = self._mid_event.cuda_event
            call.batch_struct.mid_event_late = 1 if self.mid_event_late else 0
            self._mid_event_fresh = True
        call.batch_struct.defer_join = 1 if defer_join else 0
        _lib.call("pfo_tgn_backward_ev", ctypes.byref(call.cfg), ctypes.byref(st), ctypes.byref(call.batch_struct),
                  call.ws.data_ptr(), d_emb.data_ptr(), self._flat_grad.data_ptr(), 1 if zero_first else 0, ev,
                  mean[0].data_ptr() if mean else None, int(mean[0].shape[0]) if mean else 0,
                  mean[1].data_ptr() if mean else None, _lib.stream_ptr())

    def request_zero_grad(self):
        """``optimizer.zero_grad()`` without a launch of its own: the NEXT native backward clears the flat gradient buffer
        itself (on its side stream) before accumulating; the ``.grad`` views stay attached."""
        self._zero_next = True

    @property
    def grad_split(self):
        """Element offset of the top layer's parameter block in the flat buffers (``pfo_tgn_grad_split``)."""
        if self._grad_split is None:
            v = ctypes.c_int64(0)
            _lib.call("pfo_tgn_grad_split", ctypes.byref(self._cfg), ctypes.byref(v))
            self._grad_split = int(v.value)
        return self._grad_split

    def _native_update_state(self, call, src, dst, ts, eidx):
        self._join_backward()
        self.memory._any_msg = True
        st = self._state_struct()
        _lib.call("pfo_tgn_update_state", ctypes.byref(call.cfg), ctypes.byref(st), src.data_ptr(), dst.data_ptr(),
                  ts.data_ptr(), eidx.data_ptr(), int(src.shape[0]), call.ws.data_ptr(), _lib.stream_ptr())

    # ------------------------------------------------------------------ the step
    def _assemble_roots(self, src, dst, edge_times, groups, lo, hi, R):
        """roots = [src | dst | extra groups] of this rank's shard, root_ts = the interaction's edge time per root."""
        b = hi - lo
        if b == 0:
            # empty shard (B < world): nothing to embed, but the state update still runs on every rank; one padding
            # root keeps the native call well-formed (node 0 never has neighbours; its row is dropped below)
            return (torch.zeros(1, dtype=torch.int32, device=self.device),
                    torch.full((1,), -1.0, dtype=torch.float64, device=self.device))
        roots = torch.empty(R, dtype=torch.int32, device=self.device)
        root_ts = torch.empty(R, dtype=torch.float64, device=self.device)
        ng = len(groups)
        gptr = (ctypes.c_void_p * max(1, ng))(*[t.data_ptr() for t, _ in groups])
        greps = (ctypes.c_int32 * max(1, ng))(*[r for _, r in groups])
        _lib.call("pfo_roots_assemble", src.data_ptr(), dst.data_ptr(), edge_times.data_ptr(), lo, hi, gptr, greps, ng,
                  roots.data_ptr(), root_ts.data_ptr(), _lib.stream_ptr())
        return roots, root_ts

    # ------------------------------------------------------------------ next batch's neighbourhood beside this batch's backward
    def _batch_key(self, src, dst, groups, edge_times, K):
        nf = self.neighbor_finder
        return (src.data_ptr(), dst.data_ptr(), edge_times.data_ptr(), int(src.shape[0]), int(K),
                tuple((t.data_ptr(), int(t.shape[0]), r) for t, r in groups),
                self.memory._state_version if self.use_memory else 0, id(nf), getattr(nf, "_version", 0),
                self.dp_rank, self.dp_world)

    def prefetching(self, beside_attention_backward=False):
        """Context manager: work issued inside runs on this model's prefetch stream, ordered behind everything the caller's
        stream holds at entry.  A training loop draws the NEXT batch's negatives and calls ``prefetch`` inside it, right after
        the current batch's forward: the whole preparation then runs beside the current batch's backward.

        ``beside_attention_backward`` (needs ``tgn.record_mid_event = True`` before the backward): enter AFTER the current
        batch's backward was queued; the prefetch stream then waits for
        the event that backward recorded in front of its layer-1 attention kernel (``pfo_tgn_batch.mid_event``) instead of for
        the whole backward: the preparation's small latency-bound launches run beside the longest kernel of the step, the one
        phase that hides them (beside the layer-2 backward - equally small launches - they cost as much as they save)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self._pre_stream is None:
                self._pre_stream = torch.cuda.Stream(device=self.device)
            if beside_attention_backward and self._mid_event_fresh:
                ev = self._mid_event                              # (behind the forward's state update too: same stream, earlier)
                self._mid_event_fresh = False
            else:
                ev = torch.cuda.Event()
                ev.record()                                       # after the caller's forward (whose state update the pack reads)
            self._pre_main = torch.cuda.current_stream()
            self._pre_stream.wait_event(ev)
            with torch.cuda.stream(self._pre_stream):
                yield
        return ctx()

    def prefetch(self, src, dst, extra_roots, extra_repeat, edge_times, edge_idxs, n_neighbors):
        """Everything of the next training ``embed_device`` / ``compute_temporal_embeddings[_p]`` call (SAME tensors, passed
        again there) that depends on neither parameters nor gradients: root assembly, frontier sampling (utils.py:163-219 per
        level), compaction of the touched nodes, packed copies of their memory / message rows - the work the reference's loop
        does on the host between batches.  Call inside ``with tgn.prefetching():``.  Most-recent sampling only (the random
        modes draw from the step's stream position); a call that does not match is simply not used."""
        _lib.require_gpu(self.device)
        if bool(getattr(self.neighbor_finder, "uniform", False)) or not torch.is_grad_enabled():
            return False
        self._join_backward()             # (overlap_backward: the workspace this call takes may be the one that backward reads)
        self._drop_prefetched()
        B, K = int(src.shape[0]), int(n_neighbors)
        lo, hi = 0, B
        if self.dp_world > 1:
            lo, hi = self.dp_rank * B // self.dp_world, (self.dp_rank + 1) * B // self.dp_world
        b = hi - lo
        if b == 0 or K <= 0:
            return False
        groups = [(t.contiguous(), int(r)) for t, r in zip(extra_roots, extra_repeat) if int(r) > 0]
        R = b * (2 + sum(r for _, r in groups))
        extra = torch.cat([src, dst]).contiguous() if (self.use_memory and self.dp_world > 1) else None
        roots, root_ts = self._assemble_roots(src, dst, edge_times, groups, lo, hi, R)
        dropout_p = self.dropout if self.training else 0.0
        main = self._pre_main if self._pre_main is not None else torch.cuda.default_stream(self.device)
        with torch.cuda.stream(main):
            # a NEW workspace (pool miss) must belong to the caller's stream's allocator pool: the forward / backward that use
            # it run there, and a later pool purge must free it to that stream's allocator, not the prefetch stream's
            call = self._make_call(roots, root_ts, K, None, dropout_p, extra, B, defer_step=True)
        call.batch_struct.prepared = 1
        st = self._state_struct()
        _lib.call("pfo_tgn_prepare", ctypes.byref(call.cfg), ctypes.byref(st), ctypes.byref(call.batch_struct),
                  call.ws.data_ptr(), _lib.stream_ptr())
        call.ready = torch.cuda.Event()
        call.ready.record()
        call.keep = (src, dst, edge_times, groups)                 # the key's addresses stay theirs until the call is used
        for t in (roots, root_ts, extra) + tuple(g for g, _ in groups):
            if t is not None:
                t.record_stream(main)                             # allocated on the prefetch stream, read on the caller's
        self._prefetched = (self._batch_key(src, dst, groups, edge_times, K), dropout_p, call)
        return True

    def _discard_call(self, call):
        """Hands a prepared-but-unused call's workspace back.  ``pfo_tgn_prepare`` may still be running on the prefetch
        stream: whoever pops that workspace from the pool next works on the CALLER's stream, so that stream is made to
        wait for the preparation first (stream-ordered, no host sync)."""
        ready = getattr(call, "ready", None)
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
        call.release()

    def _drop_prefetched(self):
        if self._prefetched is not None:
            self._discard_call(self._prefetched[2])
            self._prefetched = None

    def _take_prefetched(self, src, dst, groups, edge_times, K, draws, offset_dev, grad_mode, dropout_p):
        if self._prefetched is None:
            return None
        key, p_drop, call = self._prefetched
        self._prefetched = None
        if (grad_mode and draws is None and offset_dev is None and p_drop == dropout_p
                and key == self._batch_key(src, dst, groups, edge_times, K)
                and bool(self.deterministic) == bool(call.batch_struct.deterministic)):
            self._step += 1                                       # its position in the random streams (dropout masks): now
            call.offset = self._step << 36
            call.batch_struct.offset = call.offset
            return call
        self._discard_call(call)
        return None

    def embed_device(self, src, dst, extra_roots, extra_repeat, edge_times, edge_idxs, n_neighbors, draws=None, offset_dev=None,
                     dropout_keep=None, split_blocks=None):
        """Device-resident core of both reference entry points.

        src/dst i32[B], edge_times f64[B], edge_idxs i32[B], extra_roots: list of i32 tensors [B*r_k] (negatives /
        p_pos / p_neg, row-major per interaction) with repeat counts ``extra_repeat``; all on ``self.device``; node ids
        must lie in [0, n_nodes) (the numpy entry points check this, this one does not: it would cost a device sync).
        Returns the embedding matrix [R, D] in the order [src | dst | extra...] for THIS rank's shard of the batch
        (``R = b * (2 + sum(extra_repeat))``, ``b`` = the second return value; b may be 0 for a trailing rank when the
        batch is shorter than the world size), and performs the memory persist + raw-message store for the whole batch
        (tgn.py:290-317).  ``offset_dev`` (a 1-element int64 device tensor): position of the random streams for steps
        captured into a HIP graph (pfotgnrec_amd/graph.py).  ``self.dp_grad_scale`` then holds b / B, the factor that turns this shard's mean-loss
        gradient into its share of the global-batch mean gradient.
        ``split_blocks`` (block heights in units of b, e.g. (1, 1, n_neg)): the first return value is a tuple of those row
        blocks instead of the matrix - under autograd as separate outputs of one node (see _EmbedFn.forward).
        """
        _lib.require_gpu(self.device)
        B = int(src.shape[0])
        K = int(n_neighbors)
        if self.debug_checks and self.use_memory:
            m = self.memory
            late = (m.has_msg > 0) & (m.last_update > m.msg_time)
            assert not bool(late.any()), "Trying to update memory to time in the past"      # memory_updater.py:25,41
        # is any message pending?  Known on the host except right after __init_memory__ / a restore (one read-back then)
        self._gru_applied_now = True
        if self.use_memory and not self.memory._any_msg:
            self._gru_applied_now = bool(self.memory.has_msg.any())
            self.memory._any_msg = self._gru_applied_now
        lo, hi = 0, B
        if self.dp_world > 1:
            lo, hi = self.dp_rank * B // self.dp_world, (self.dp_rank + 1) * B // self.dp_world
        b = hi - lo
        self.dp_grad_scale = (b / float(B)) if B > 0 else 0.0
        # roots = [src | dst | extra groups], root_ts = the interaction's edge time per root (tgn.py:123-124 / 238-239);
        # a group with zero nodes per interaction (e.g. no negatives) contributes nothing
        groups = [(t.contiguous(), int(r)) for t, r in zip(extra_roots, extra_repeat) if int(r) > 0]
        R = b * (2 + sum(r for _, r in groups))
        D = self.n_node_features
        grad_mode = torch.is_grad_enabled()
        dropout_p = self.dropout if self.training else 0.0       # dropout follows train()/eval(), not the autograd mode
        if dropout_keep is not None:
            self._drop_prefetched()
        pre = self._take_prefetched(src, dst, groups, edge_times, K, draws, offset_dev, grad_mode, dropout_p)
        if pre is not None:
            roots, root_ts, extra = pre.roots, pre.root_ts, pre.extra
        else:
            extra = torch.cat([src, dst]).contiguous() if (self.use_memory and self.dp_world > 1) else None
            roots, root_ts = self._assemble_roots(src, dst, edge_times, groups, lo, hi, R)
        no_neighbours = K <= 0
        if no_neighbours:                # utils.py:175: a single all-padding column
            K, root_ts = 1, torch.full_like(root_ts, -1.0)
        post = None
        if self.use_memory:
            post = lambda call: self._native_update_state(call, src, dst, edge_times, edge_idxs)
        fuse_state = self.use_memory and self.n_layers >= 2 and b > 0 and grad_mode and self.fuse_state_update
        if b == 0:
            call = self._make_call(roots, root_ts, K, None, 0.0, extra, B)
            self._native_forward(call)
            if post is not None:
                post(call)
            call.release()
            if grad_mode:
                self._attach_grads(self._gru_applied_now)         # this rank still joins the all-reduce: with a zero gradient
                if self._zero_next:                               # request_zero_grad(): no backward will run to honour it
                    self._flat_grad.zero_()
                    self._zero_next = False
                self._bucket_event_fresh = False                  # no backward ran to record the event (the collectives stay the same)
            # a leaf that requires grad: the caller's loss.backward() is a no-op instead of an error
            z = torch.zeros((0, D), dtype=torch.float32, device=self.device, requires_grad=grad_mode)
            return (z if split_blocks is None else tuple(z[0:0] for _ in split_blocks)), 0
        split = None
        if split_blocks is not None:
            edges = np.concatenate([[0], np.cumsum(split_blocks)]) * b
            split = tuple((int(r0), int(r1)) for r0, r1 in zip(edges[:-1], edges[1:]))
            assert split[-1][1] == R, "split_blocks must cover the roots: (1, 1, repeat counts of the extra groups)"
        if grad_mode:
            if pre is not None:
                call = pre
                call.gru_applied = self._gru_applied_now
                torch.cuda.current_stream().wait_event(call.ready)    # the frontier, the compaction and the packed rows exist
            else:
                call = self._make_call(roots, root_ts, K, draws, dropout_p, extra, B, offset_dev, dropout_keep=dropout_keep)
            if fuse_state:
                # the native forward performs the state update itself (pfo_tgn_batch.upd_*): on its side stream, beside layer 1
                bs = call.batch_struct
                bs.upd_src, bs.upd_dst, bs.upd_ts, bs.upd_eidx, bs.upd_B = (src.data_ptr(), dst.data_ptr(), edge_times.data_ptr(),
                                                                           edge_idxs.data_ptr(), B)
                # (one rank only: a data-parallel rank's forward-side stream already carries the GLOBAL batch's state update -
                #  8x the work at 8 ranks - and the event layer 2 waits for would move behind it: emulated 8-rank step 1.547 ms
                #  with, 1.537 without)
                bs.seg_in_forward = 1 if (self.seg_in_forward and (self.dp_world == 1 or self.seg_in_forward_dp)) else 0
                call.keep = (getattr(call, "keep", None), src, dst, edge_times, edge_idxs)
                self.memory._any_msg = True
                post = None
            # (ONE parameter that requires a gradient ties the node into the graph: the native backward accumulates every
            #  parameter gradient itself, and autograd's per-input bookkeeping of ~50 tensors was ~15 us of host time per batch)
            anchor = next((v[0] for v in self._views if v[0].requires_grad), None)
            emb = _EmbedFn.apply(self, call, post, split, *(() if anchor is None else (anchor,)))
            return emb, b
        # forward only (evaluation.py:94: R = B*(2+N_ITEMS) roots): walk the roots in chunks through the same
        # kernels; memory is persisted once, after the last chunk, from a pass that covers the positives.
        # An embedding is a function of (node, time) only, and every interaction of an evaluation batch scores the SAME
        # items (evaluation.py:88-89): interactions that share a timestamp (day-granular data) repeat whole blocks of
        # roots.  Those are embedded once and gathered back.
        inverse = None
        if self.eval_dedup and draws is None and R >= 4096 and not no_neighbours:   # (the grid is rebuilt from the real edge
            # times: with the K <= 0 override every root's time is -1 and the list is embedded as it stands)
            roots, root_ts, inverse = self._dedup_roots(src[lo:hi], dst[lo:hi], edge_times[lo:hi], groups, lo, hi, B, roots, root_ts)
            R = int(roots.shape[0])
        cap = int(self.eval_chunk_roots)
        if dropout_keep is not None and (inverse is not None or R > int(self.eval_chunk_roots)):
            raise ValueError("injected dropout decisions address the roots as given: not with the forward-only dedup / chunk walk")
        if R <= cap or draws is not None:
            call = self._make_call(roots, root_ts, K, draws, dropout_p, extra, B, dropout_keep=dropout_keep)
            emb = self._native_forward(call)
        else:
            emb = torch.empty((R, D), dtype=torch.float32, device=self.device)
            pos = torch.cat([src, dst]).contiguous() if self.use_memory else None
            order = list(range(cap, R, cap)) + [0]                # chunk 0 (holds src|dst) last: its touched set feeds the persist
            call = None
            for c0 in order:
                c1 = min(R, c0 + cap)
                ex = extra if c0 != 0 or pos is None else (pos if extra is None else extra)
                if call is not None:
                    call.release()
                call = self._make_call(roots[c0:c1], root_ts[c0:c1], K, None, dropout_p, ex, B)
                self._native_forward(call, out=emb[c0:c1])
        if post is not None:
            post(call)
        call.release()
        if inverse is not None:
            emb = emb.index_select(0, inverse)
        if split is not None:
            emb = tuple(emb[r0:r1] for r0, r1 in split)
        return emb, b

    def _dedup_roots(self, src, dst, ts, groups, lo, hi, B, roots, root_ts):
        """An embedding is a function of (node, time) only, and an evaluation batch scores the same few hundred items for
        every interaction (evaluation.py:88-94: N_ITEMS draws from the item set per interaction, all at the interaction's
        time).  Item-side structure: the distinct timestamps T of the batch (a sort of b values) and the
        distinct nodes U of the extra groups (a presence bitmap over the node table, no sort) span a dense grid of T x U
        roots; every extra root is a gather from it.  Taken when the grid is smaller than the list it replaces.
        Returns (roots, root_ts, inverse) with ``inverse`` = None when nothing is saved."""
        b = hi - lo
        n_extra = sum(r for _, r in groups)
        if b == 0 or n_extra == 0:
            return roots, root_ts, None
        uniq_t, tid = torch.unique(ts, return_inverse=True)                       # b values
        present = torch.zeros(self.n_nodes, dtype=torch.bool, device=self.device)
        cols = [t.view(B, r)[lo:hi] for t, r in groups]                             # this shard's rows of every group
        for c in cols:
            present[c.reshape(-1).long()] = True
        uniq_n = present.nonzero().view(-1)
        n_t, n_u = int(uniq_t.shape[0]), int(uniq_n.shape[0])                      # (one host read-back for both)
        if n_t * n_u >= b * n_extra:
            return roots, root_ts, None
        col_of = torch.cumsum(present, 0, dtype=torch.int32) - 1                    # node id -> column of the grid
        grid_nodes = uniq_n.to(torch.int32).repeat(n_t)
        grid_ts = uniq_t.repeat_interleave(n_u)
        new_roots = torch.cat([roots[:2 * b], grid_nodes]).contiguous()
        new_ts = torch.cat([root_ts[:2 * b], grid_ts]).contiguous()
        base = 2 * b + tid.to(torch.int64) * n_u                                     # [b] first grid row of the interaction's time
        inv = [torch.arange(2 * b, device=self.device, dtype=torch.int64)]
        for c in cols:
            inv.append((base[:, None] + col_of[c.long()].to(torch.int64)).reshape(-1))
        return new_roots, new_ts, torch.cat(inv)

    def train(self, mode=True):
        # main.py calls ``tgn = tgn.train()`` in front of EVERY batch (main.py:308,354): nn.Module.train walks all submodules and
        # sets an attribute on each - 0.1 ms per batch on the drop-in loop for a mode that does not change
        # (short-circuit only while EVERY submodule already is in that mode: a child toggled on its own - tgn.child.eval() - or
        #  attached after the last call is reset exactly as nn.Module.train would; the walk without the attribute writes is 5 us)
        mode = bool(mode)
        if self.training == mode:
            # (the module LIST is kept: walking named_modules() is 25 us per batch; a module attached to this model later
            #  drops it - __setattr__ / add_module below - and a tree edited underneath a child is caught by the count)
            mods = self.__dict__.get("_module_list")
            if mods is None or len(mods[0]._modules) != mods[1]:
                ms = list(self.modules())
                mods = self.__dict__["_module_list"] = (self, len(self._modules), ms, sum(len(m._modules) for m in ms))
            if all(m.training == mode for m in mods[2]) and sum(len(m._modules) for m in mods[2]) == mods[3]:
                return self
            self.__dict__["_module_list"] = None
        return super().train(mode)

    def __setattr__(self, name, value):
        # plain private state (step counters, cache keys, events ...) is written a dozen times per step: nn.Module.__setattr__
        # walks its parameter / buffer / module tables first (~3 us each).  Parameters and modules take the usual route.
        if name[0] == "_" and not isinstance(value, (nn.Parameter, nn.Module)):
            object.__setattr__(self, name, value)
            return
        if isinstance(value, nn.Module):
            self.__dict__["_module_list"] = None
        super().__setattr__(name, value)

    def add_module(self, name, module):
        self.__dict__["_module_list"] = None
        return super().add_module(name, module)

    def _batch_to_dev(self, parts):
        """Several small host arrays of one batch in ONE host-to-device copy: ``parts`` = [(array, numpy dtype)], widest dtype
        first; returns the device tensors in the same order (five separate copies cost the drop-in loop ~60 us of host time).
        The arrays are converted straight into one of two page-locked staging buffers (an event per buffer says when its last
        copy has left it) and the copy is asynchronous on the caller's stream: the host goes on to build the call while it
        runs (a pageable copy holds the host for ~25 us with the device idle behind it)."""
        _lib.require_gpu(self.device)
        arrs = [np.asarray(a) for a, _ in parts]
        dts = [np.dtype(dt) for _, dt in parts]
        sizes = [a.size * dt.itemsize for a, dt in zip(arrs, dts)]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 7) & ~7
        ring = self.__dict__.setdefault("_stage_ring", [])
        slot = self.__dict__["_stage_next"] = (self.__dict__.get("_stage_next", -1) + 1) % 2
        if len(ring) <= slot or ring[slot][0].numel() < total:
            host = torch.empty(max(total, 1 << 16), dtype=torch.uint8, pin_memory=True)
            entry = (host, host.numpy(), torch.cuda.Event())
            if len(ring) <= slot:
                ring.append(entry)
            else:
                ring[slot][2].synchronize()
                ring[slot] = entry
        else:
            ring[slot][2].synchronize()                       # (its last copy: two batches ago - normally long done)
        host, view, ev = ring[slot]
        for a, dt, o, n in zip(arrs, dts, offs, sizes):
            view[o:o + n].view(dt)[...] = a.reshape(-1)       # converts while it copies
        dev = host[:total].to(self.device, non_blocking=True)
        ev.record()
        out = []
        for a, dt, o, n in zip(arrs, dts, offs, sizes):
            out.append(dev[o:o + n].view(_TORCH_OF[dt.name]).view(a.shape))
        return out

    def _check_nodes(self, a, what):
        a = np.asarray(a)
        if a.size and (int(a.min()) < 0 or int(a.max()) >= self.n_nodes):
            raise IndexError("%s holds node ids outside [0, %d)" % (what, self.n_nodes))
        return a

    def _check_edges(self, a):
        a = np.asarray(a)
        if a.size and (int(a.min()) < 0 or int(a.max()) >= self.edge_raw_features.shape[0]):
            raise IndexError("edge_idxs outside [0, %d)" % self.edge_raw_features.shape[0])
        return a

    def _to_dev(self, a, dtype):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dtype)).to(self.device)

    def _nodes_to_dev(self, a, what):
        """Host node ids -> i32 device tensor; ids outside [0, n_nodes) raise IndexError like the reference's table
        lookups would (the native kernels index feature / memory / adjacency tables with them unchecked)."""
        a = np.asarray(a)
        if a.size and (int(a.min()) < 0 or int(a.max()) >= self.n_nodes):
            raise IndexError("%s holds node ids outside [0, %d)" % (what, self.n_nodes))
        return self._to_dev(a, np.int32)

    def _edges_to_dev(self, a):
        a = np.asarray(a)
        if a.size and (int(a.min()) < 0 or int(a.max()) >= self.edge_raw_features.shape[0]):
            raise IndexError("edge_idxs outside [0, %d)" % self.edge_raw_features.shape[0])
        return self._to_dev(a, np.int32)

    def compute_temporal_embeddings(self, source_nodes, destination_nodes, p_neg_nodes, edge_times, edge_idxs,
                                    n_neighbors=20, draws=None, dropout_keep=None):
        """tgn.py:219-327.  numpy in (i64, i64, i64 flat row-major, f64, i64), device tensors out:
        (src_emb [B,D], dst_emb [B,D], neg_emb [B*size,D])."""
        B = len(source_nodes)
        size = int(len(p_neg_nodes) / B)                                        # tgn.py:237
        ts, src, dst, neg, eidx = self._batch_to_dev([(edge_times, np.float64), (self._check_nodes(source_nodes, "source_nodes"), np.int32),
                                                      (self._check_nodes(destination_nodes, "destination_nodes"), np.int32),
                                                      (self._check_nodes(p_neg_nodes, "p_neg_nodes"), np.int32),
                                                      (self._check_edges(edge_idxs), np.int32)])
        out, _ = self.embed_device(src, dst, [neg], [size], ts, eidx, n_neighbors, self._dev_draws(draws),
                                   dropout_keep=self._dev_keep(dropout_keep), split_blocks=(1, 1, size))
        return out

    def compute_temporal_embeddings_p(self, source_nodes, destination_nodes, p_pos_nodes, p_neg_nodes, edge_times,
                                      edge_idxs, n_neighbors=20, draws=None):
        """tgn.py:102-217: (src_emb, dst_emb, p_pos_emb [B*p,D], p_neg_emb [B*q,D])."""
        B = len(source_nodes)
        n_pos, n_neg = int(len(p_pos_nodes) / B), int(len(p_neg_nodes) / B)           # tgn.py:118-119
        ts, src, dst, pp, pn, eidx = self._batch_to_dev([(edge_times, np.float64), (self._check_nodes(source_nodes, "source_nodes"), np.int32),
                                                         (self._check_nodes(destination_nodes, "destination_nodes"), np.int32),
                                                         (self._check_nodes(p_pos_nodes, "p_pos_nodes"), np.int32),
                                                         (self._check_nodes(p_neg_nodes, "p_neg_nodes"), np.int32),
                                                         (self._check_edges(edge_idxs), np.int32)])
        out, _ = self.embed_device(src, dst, [pp, pn], [n_pos, n_neg], ts, eidx, n_neighbors, self._dev_draws(draws),
                                   split_blocks=(1, 1, n_pos, n_neg))
        return out

    def _dev_keep(self, masks):
        """Injected dropout decisions for a parity test: ``{l: multipliers or booleans [n_l, H, K]}`` (level order, the layout
        of ``debug_dropout_masks`` and of the oracle's ``dropout_masks``) -> per layer, roots' level first, u8 [n_l, K] with bit
        h = head h kept."""
        if masks is None:
            return None
        out = []
        for l in range(self.n_layers, 0, -1):
            m = np.asarray(masks[l]) != 0                                   # [n_l, H, K]
            bits = np.zeros((m.shape[0], m.shape[2]), np.uint8)
            for h in range(m.shape[1]):
                bits |= (m[:, h, :].astype(np.uint8) << h)
            out.append(torch.from_numpy(np.ascontiguousarray(bits)).to(self.device))
        return out

    def _dev_draws(self, draws):
        if draws is None:
            return None
        return [self._to_dev(d, np.int64) for d in draws]

    # ------------------------------------------------------------------ introspection for tests
    def debug_dropout_masks(self):
        """The train-mode dropout multipliers the LAST forward applied to its attention weights, per layer:
        ``{l: f32 ndarray [n_l, H, K]}`` in level order (n_L = R roots, n_{l-1} = n_l (1 + K)); 1/(1-p) = kept, 0 = dropped.
        Regenerated through the C ABI from the call's seed and stream position (``pfo_attn_dropout_mask``): what an oracle
        needs to replay the step with the same masks.  None when the call ran without dropout."""
        seed, offset, R, K, p, dev_offset = self._last_call
        if p <= 0.0:
            return None
        if dev_offset:
            raise RuntimeError("the stream position of a graph-captured step lives on the device")
        out, n = {}, R
        for l in range(self.n_layers, 0, -1):
            m = torch.empty((n, self.n_heads, K), dtype=torch.float32, device=self.device)
            _lib.call("pfo_attn_dropout_mask", seed, offset + 0x51ED0000 + l, n, K, self.n_heads, p, m.data_ptr(), _lib.stream_ptr())
            out[l] = m.cpu().numpy()
            n *= 1 + K
        return out

    def debug_weight_gradient_operands(self):
        """fp32 operands and results of the two largest weight-gradient contractions of the LAST backward (before the next
        forward reuses the workspace): ``dict(ctx [n_1, H*Cp], dh1 [n_1, D], dW1ovT [H*Cp, D], dgi [n_core, 3D],
        msg_rows [n_core, 3D+Ef])`` as numpy arrays (tests re-contract them in fp64)."""
        cfg, ws = self._last_ws
        seed, offset, R, K, p, _ = self._last_call
        dbg = _lib.TgnDebug()
        _lib.call("pfo_tgn_debug_views", ctypes.byref(cfg), ws.data_ptr(), ctypes.byref(dbg))
        base = ws.data_ptr()

        def view(ptr, count, dtype=torch.float32):
            off = ptr - base
            return ws[off:off + count * torch.empty((), dtype=dtype).element_size()].view(dtype)
        D, H, Cp = self.n_node_features, self.n_heads, int(dbg.Cp)
        n1 = R * (1 + K) ** (self.n_layers - 1)
        out = dict(ctx=view(dbg.l1_ctx, n1 * H * Cp).view(n1, H * Cp).cpu().numpy(),
                   dh1=view(dbg.l1_dh1, n1 * D).view(n1, D).cpu().numpy(),
                   dW1ovT=view(dbg.l1_dW1ovT, H * Cp * D).view(H * Cp, D).cpu().numpy())
        if self.use_memory:
            nc = int(view(dbg.n_core, 1, torch.int32).item())
            M = 3 * D + self.n_edge_features
            out.update(dgi=view(dbg.gru_dgi, nc * 3 * D).view(nc, 3 * D).cpu().numpy(),
                       msg_rows=view(dbg.gru_msg_rows, nc * M).view(nc, M).cpu().numpy())
        return out

    def debug_touched(self):
        """(touched node ids, layer-0 feature table rows) of the last forward (use_memory only)."""
        cfg, ws = self._last_ws
        dbg = _lib.TgnDebug()
        _lib.call("pfo_tgn_debug_views", ctypes.byref(cfg), ws.data_ptr(), ctypes.byref(dbg))
        base = ws.data_ptr()

        def view(ptr, count, dtype):
            off = ptr - base
            return ws[off:off + count * torch.empty((), dtype=dtype).element_size()].view(dtype)
        n = int(view(dbg.n_touched, 1, torch.int32).item())
        ids = view(dbg.touched_ids, n, torch.int32).clone()
        h0 = view(dbg.h0_table, n * self.n_node_features, torch.float32).view(n, -1).clone()
        return ids, h0
