"""Oracle: TGN forward/backward + memory state machine in numpy fp32.  Test infrastructure only.

Restates, in the reference's own operation order (un-folded K/V projections,
materialised key tensor, recursive embedding), what these reference files
compute on the training path:

* model/time_encoding.py:17-25        -> ``time_encode``
* modules/memory_updater.py:18-61     -> ``gru_cell`` / ``_get_updated_memory`` / ``_update_memory``
* modules/message_aggregator.py:38-55 -> ``_aggregate_last``
* model/temporal_attention.py:34-90   -> ``attention_forward`` / ``attention_backward``
* utils/utils.py:4-17 (MergeLayer)    -> inside the two attention functions
* modules/embedding_module.py:76-175  -> ``_embed`` / ``_embed_backward`` (recursion)
* model/tgn.py:102-378                -> ``compute_temporal_embeddings[_p]``, ``_get_raw_messages``
* main.py:321-337 / 364-381           -> ``bpr_loss`` / ``bpr_loss_backward``

Parameters are held in a dict keyed by the reference's ``state_dict`` names so
that golden fixtures captured from the reference inject directly.  Dropout is
not modelled (parity is defined at p=0 / eval mode).
Backward is hand-derived; it is pinned against the reference's autograd by the
golden fixtures (tests/test_oracle_golden.py).
"""
from collections import defaultdict
import numpy as np

f32 = np.float32


# ----------------------------------------------------------------------------- exact fp32 FMA
def fmaf(a, b, c):
    """float32 fused multiply-add with a single rounding (what torch-CPU ``Linear(1, D)`` does).

    a*b is exact in f64 (24+24 <= 53 bits); the f64 sum is rounded to odd so the
    final f64->f32 rounding is the only effective one.
    """
    a64 = np.asarray(a, f32).astype(np.float64)
    b64 = np.asarray(b, f32).astype(np.float64)
    c64 = np.asarray(c, f32).astype(np.float64)
    p = a64 * b64
    p, c64 = np.broadcast_arrays(p, c64)
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)                      # TwoSum error term (exact)
    bits = s.view(np.int64)
    need = (err != 0) & ((bits & 1) == 0) & np.isfinite(s)
    toward = np.where(err > 0, np.inf, -np.inf)
    s = np.where(need, np.nextafter(s, toward), s)
    return s.astype(f32)


_CLIB = None


def _clib():
    """oracle/_build/liboracle.so (C99 fmaf + cosf, built by `make -C oracle`); None -> numpy emulation."""
    global _CLIB
    if _CLIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle.so")
        _CLIB = False
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
            lib.oracle_time_encode.argtypes = [vp, i64, vp, vp, i32, vp]
            lib.oracle_time_encode_bwd.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp]
            _CLIB = lib
    return _CLIB or None


def time_encode(t, w, b):
    """cos(fma(t, w_d, b_d)); t f32[...]; w [D] (or [D,1]); returns f32[..., D]  (time_encoding.py:17-25)."""
    lib = _clib()
    w1, b1 = np.ascontiguousarray(w, f32).reshape(-1), np.ascontiguousarray(b, f32)
    if lib is not None and len(w1) <= 512:
        tt = np.ascontiguousarray(t, f32)
        out = np.empty(tt.shape + (len(w1),), f32)
        lib.oracle_time_encode(tt.ctypes.data, tt.size, w1.ctypes.data, b1.ctypes.data, len(w1), out.ctypes.data)
        return out
    t = np.asarray(t, f32)[..., None]
    return np.cos(fmaf(t, w1, b1)).astype(f32)


def time_encode_backward(t, w, b, g):
    """Grads of sum(g * cos(t*w+b)) w.r.t. w [D] and b [D]."""
    lib = _clib()
    w1, b1 = np.ascontiguousarray(w, f32).reshape(-1), np.ascontiguousarray(b, f32)
    if lib is not None and len(w1) <= 512:
        tt, gg = np.ascontiguousarray(t, f32), np.ascontiguousarray(g, f32)
        dw, db = np.empty(len(w1), np.float64), np.empty(len(w1), np.float64)
        lib.oracle_time_encode_bwd(tt.ctypes.data, tt.size, w1.ctypes.data, b1.ctypes.data, len(w1), gg.ctypes.data,
                                   dw.ctypes.data, db.ctypes.data)
        return dw.astype(f32), db.astype(f32)
    t = np.asarray(t, f32)[..., None]
    s = -np.sin(fmaf(t, np.asarray(w, f32).reshape(-1), np.asarray(b, f32))).astype(f32) * g
    D = s.shape[-1]
    return (s * t).reshape(-1, D).sum(0, dtype=np.float64).astype(f32), s.reshape(-1, D).sum(0, dtype=np.float64).astype(f32)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)


# ----------------------------------------------------------------------------- GRU cell
def gru_cell(x, h, W_ih, W_hh, b_ih, b_hh):
    """torch.nn.GRUCell (memory_updater.py:60): gates ordered (r, z, n)."""
    D = h.shape[1]
    gi = x @ W_ih.T + b_ih
    gh = h @ W_hh.T + b_hh
    r = sigmoid(gi[:, :D] + gh[:, :D])
    z = sigmoid(gi[:, D:2 * D] + gh[:, D:2 * D])
    n = np.tanh(gi[:, 2 * D:] + r * gh[:, 2 * D:]).astype(f32)
    hn = ((1 - z) * n + z * h).astype(f32)
    return hn, (x, h, r, z, n, gh[:, 2 * D:])


def gru_cell_backward(cache, d_hn, W_ih, W_hh):
    """Parameter grads only: x (stored message) and h (detached memory) are constants."""
    x, h, r, z, n, ghn = cache
    dn = d_hn * (1 - z)
    dz = d_hn * (h - n)
    dpre_n = dn * (1 - n * n)
    dr = dpre_n * ghn
    dpre_r = dr * r * (1 - r)
    dpre_z = dz * z * (1 - z)
    dgi = np.concatenate([dpre_r, dpre_z, dpre_n], 1).astype(f32)
    dgh = np.concatenate([dpre_r, dpre_z, dpre_n * r], 1).astype(f32)
    return {"weight_ih": dgi.T @ x, "weight_hh": dgh.T @ h, "bias_ih": dgi.sum(0), "bias_hh": dgh.sum(0)}


# ----------------------------------------------------------------------------- temporal attention layer
def attention_forward(p, x, tq, nbr_feat, ef, te, mask, n_head, drop=None):
    """TemporalAttentionLayer.forward (temporal_attention.py:34-90).

    p: dict with Wq [E,E], Wk [E,C], Wv [E,C], b_in [3E], Wo [E,E], bo [E], W1 [D,E+D], b1, W2 [D,D], b2
    x [N,D]; tq [N,D]; nbr_feat [N,K,D]; ef [N,K,Ef]; te [N,K,D]; mask bool [N,K] (True = padding).
    drop: None (dropout 0 / eval mode) or f32 [N,H,K], the multiplier nn.MultiheadAttention's train-mode dropout puts on the
    softmax weights (temporal_attention.py:28,70: ``dropout=dropout`` -> F.dropout(attn_output_weights, p) before the
    product with V): 1/(1-p) where the weight is kept, 0 where it is dropped.  The mask itself is INJECTED (the RNG stream
    that draws it is not part of the contract).
    """
    N, K = mask.shape
    q_in = np.concatenate([x, tq], 1)                                  # :51
    key = np.concatenate([nbr_feat, ef, te], 2)                        # :52 (neighbour, edge, time)
    E = q_in.shape[1]
    inv = mask.all(1)                                                  # :60
    mask = mask.copy()
    mask[inv, 0] = False                                               # :65
    bq, bk, bv = p["b_in"][:E], p["b_in"][E:2 * E], p["b_in"][2 * E:]
    Qp = q_in @ p["Wq"].T + bq
    key2 = key.reshape(N * K, -1)                                      # one large GEMM instead of N small ones
    Kp = (key2 @ p["Wk"].T + bk).reshape(N, K, E)                      # [N,K,E]
    Vp = (key2 @ p["Wv"].T + bv).reshape(N, K, E)
    dh = E // n_head
    scale = f32(1.0 / np.sqrt(dh))
    Qh = (Qp * scale).reshape(N, n_head, dh)
    Kh = Kp.reshape(N, K, n_head, dh)
    Vh = Vp.reshape(N, K, n_head, dh)
    scores = np.matmul(Kh.transpose(0, 2, 1, 3), Qh[:, :, :, None])[..., 0]          # [N,H,K] (batched BLAS)
    scores = np.where(mask[:, None, :], -np.inf, scores).astype(f32)
    m = scores.max(-1, keepdims=True)
    e = np.exp(scores - m)
    a = (e / e.sum(-1, keepdims=True)).astype(f32)                     # [N,H,K]
    ad = a if drop is None else (a * drop).astype(f32)                 # F.dropout on the attention weights (train mode)
    Oh = np.matmul(ad[:, :, None, :], Vh.transpose(0, 2, 1, 3))[:, :, 0, :].reshape(N, E)
    attn = Oh @ p["Wo"].T + p["bo"]
    attn[inv] = 0                                                      # :84
    cat = np.concatenate([attn, x], 1)                                 # :88 / utils.py:15
    z1 = cat @ p["W1"].T + p["b1"]
    h1 = np.maximum(z1, 0)
    out = (h1 @ p["W2"].T + p["b2"]).astype(f32)
    cache = dict(q_in=q_in, key=key, inv=inv, Qh=Qh, Kh=Kh, Vh=Vh, a=a, ad=ad, drop=drop, Oh=Oh, cat=cat, z1=z1, h1=h1, scale=scale)
    return out, cache


def attention_backward(p, c, d_out, n_head, D):
    """Returns (param grads, d_x [N,D], d_tq [N,D], d_nbr_feat [N,K,D], d_te [N,K,D])."""
    N, K = c["a"].shape[0], c["a"].shape[2]
    E = c["q_in"].shape[1]
    g = {}
    g["W2"] = d_out.T @ c["h1"]; g["b2"] = d_out.sum(0)
    dh1 = d_out @ p["W2"]
    dz1 = dh1 * (c["z1"] > 0)
    g["W1"] = dz1.T @ c["cat"]; g["b1"] = dz1.sum(0)
    dcat = dz1 @ p["W1"]
    dattn = dcat[:, :E].copy()
    d_x = dcat[:, E:].copy()
    dattn[c["inv"]] = 0
    g["Wo"] = dattn.T @ c["Oh"]; g["bo"] = dattn.sum(0)
    dOh = (dattn @ p["Wo"]).reshape(N, n_head, E // n_head)
    da = np.matmul(c["Vh"].transpose(0, 2, 1, 3), dOh[:, :, :, None])[..., 0]      # d loss / d (post-dropout weight)
    dVh = (c["ad"].transpose(0, 2, 1)[:, :, :, None] * dOh[:, None, :, :])
    if c["drop"] is not None:
        da = da * c["drop"]                                             # through the dropout multiplier
    a = c["a"]
    ds = a * (da - (a * da).sum(-1, keepdims=True))                     # softmax backward; masked a == 0
    dQh = np.matmul(ds[:, :, None, :], c["Kh"].transpose(0, 2, 1, 3))[:, :, 0, :]
    dKh = (ds.transpose(0, 2, 1)[:, :, :, None] * c["Qh"][:, None, :, :])
    dQp = (dQh * c["scale"]).reshape(N, E)
    dKp = dKh.reshape(N * K, E)
    dVp = dVh.reshape(N * K, E)
    key2 = c["key"].reshape(N * K, -1)
    g["Wq"] = dQp.T @ c["q_in"]
    g["Wk"] = dKp.T @ key2
    g["Wv"] = dVp.T @ key2
    g["b_in"] = np.concatenate([dQp.sum(0), dKp.sum(0), dVp.sum(0)])
    dq_in = dQp @ p["Wq"]
    dkey = (dKp @ p["Wk"] + dVp @ p["Wv"]).reshape(N, K, -1)
    d_x += dq_in[:, :D]
    d_tq = dq_in[:, D:]
    Ef = dkey.shape[2] - 2 * D
    return ({k: v.astype(f32) for k, v in g.items()}, d_x.astype(f32), d_tq.astype(f32),
            dkey[:, :, :D].astype(f32), dkey[:, :, D + Ef:].astype(f32))


# ----------------------------------------------------------------------------- the same two functions over row chunks, threaded
# The layer is row-wise independent over its N instances, and at the benchmark's size (53 760 instances x 20 neighbours) the
# un-chunked functions spend most of their time in single-threaded numpy element-wise passes over [N,K,E] arrays while all but
# one core idle.  Above MT_MIN_ROWS instances the rows are cut into chunks handled by a thread pool (numpy releases the GIL in
# ufuncs, copies and BLAS calls); parameter gradients are summed in chunk order, so results do not depend on thread timing.
# Small calls (every golden-fixture check) take the plain functions: bit-identical to before.
MT_MIN_ROWS = 8192
MT_CHUNK = 2048
_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(64, os.cpu_count() or 1)))
    return _POOL


class _one_blas_thread:
    """While the pool's workers run, every BLAS call is single-threaded: the parallelism is over the row chunks (one chunk's
    GEMMs on all cores at once, from dozens of Python threads, oversubscribes the host 8-64x)."""

    def __enter__(self):
        try:
            from threadpoolctl import threadpool_limits
            self._ctx = threadpool_limits(limits=1, user_api="blas")
            self._ctx.__enter__()
        except Exception:
            self._ctx = None

    def __exit__(self, *a):
        if self._ctx is not None:
            self._ctx.__exit__(*a)


def attention_forward_mt(p, x, tq, nbr_feat, ef, te, mask, n_head, drop=None):
    N = mask.shape[0]
    if N < MT_MIN_ROWS:
        return attention_forward(p, x, tq, nbr_feat, ef, te, mask, n_head, drop)
    bounds = [(i, min(N, i + MT_CHUNK)) for i in range(0, N, MT_CHUNK)]
    with _one_blas_thread():
        res = list(_pool().map(lambda b: attention_forward(p, x[b[0]:b[1]], tq[b[0]:b[1]], nbr_feat[b[0]:b[1]], ef[b[0]:b[1]],
                                                           te[b[0]:b[1]], mask[b[0]:b[1]], n_head,
                                                           None if drop is None else drop[b[0]:b[1]]), bounds))
    out = np.concatenate([r[0] for r in res])
    cache = dict(chunks=[r[1] for r in res], bounds=bounds, z1=np.concatenate([r[1]["z1"] for r in res]))
    return out, cache


def attention_backward_mt(p, c, d_out, n_head, D):
    if "chunks" not in c:
        return attention_backward(p, c, d_out, n_head, D)
    with _one_blas_thread():
        res = list(_pool().map(lambda cb: attention_backward(p, cb[0], d_out[cb[1][0]:cb[1][1]], n_head, D), zip(c["chunks"], c["bounds"])))
    g = {k: v.copy() for k, v in res[0][0].items()}
    for r in res[1:]:
        for k, v in r[0].items():
            g[k] += v
    return (g,) + tuple(np.concatenate([r[i] for r in res]) for i in range(1, 5))


# ----------------------------------------------------------------------------- BPR (main.py:321-337)
def bpr_loss(src, pos, neg):
    """src [B,D]; pos [B,p,D]; neg [B,q,D].  sigma of the MEAN difference."""
    pos_s = np.einsum("bd,bpd->bp", src, pos)                          # [B,p]  (p == 1 in the reference)
    neg_s = np.einsum("bd,bqd->bq", src, neg)
    diff = pos_s - neg_s                                               # broadcast [B,1]-[B,q]
    dm = diff.mean(1)
    sg = sigmoid(dm)
    loss = -np.mean(np.log(sg))
    return f32(loss), (src, pos, neg, sg, diff.shape[1])


def bpr_loss_backward(cache):
    src, pos, neg, sg, q = cache
    B = src.shape[0]
    ddm = -(1 - sg) / B                                                # d loss / d dm
    ddiff = np.repeat((ddm / q)[:, None], q, 1)                        # [B,q]
    dpos_s = ddiff.sum(1, keepdims=True) if pos.shape[1] == 1 else ddiff
    dneg_s = -ddiff
    d_src = np.einsum("bp,bpd->bd", dpos_s, pos) + np.einsum("bq,bqd->bd", dneg_s, neg)
    d_pos = dpos_s[:, :, None] * src[:, None, :]
    d_neg = dneg_s[:, :, None] * src[:, None, :]
    return d_src.astype(f32), d_pos.astype(f32), d_neg.astype(f32)


# ----------------------------------------------------------------------------- the model
def normalise_edge_features(edge_features):
    """tgn.py:38-41 - fp32 z-score per column, padding row 0 included."""
    ef = np.asarray(edge_features).astype(f32)
    ef -= ef.mean(axis=0)
    ef /= ef.std(axis=0)
    return ef.astype(f32)


def layer_params(P, l):
    pre = "embedding_module.attention_models.%d." % l
    return dict(Wq=P[pre + "multi_head_target.q_proj_weight"], Wk=P[pre + "multi_head_target.k_proj_weight"],
                Wv=P[pre + "multi_head_target.v_proj_weight"], b_in=P[pre + "multi_head_target.in_proj_bias"],
                Wo=P[pre + "multi_head_target.out_proj.weight"], bo=P[pre + "multi_head_target.out_proj.bias"],
                W1=P[pre + "merger.fc1.weight"], b1=P[pre + "merger.fc1.bias"],
                W2=P[pre + "merger.fc2.weight"], b2=P[pre + "merger.fc2.bias"])


_LAYER_KEYS = dict(Wq="multi_head_target.q_proj_weight", Wk="multi_head_target.k_proj_weight",
                   Wv="multi_head_target.v_proj_weight", b_in="multi_head_target.in_proj_bias",
                   Wo="multi_head_target.out_proj.weight", bo="multi_head_target.out_proj.bias",
                   W1="merger.fc1.weight", b1="merger.fc1.bias", W2="merger.fc2.weight", b2="merger.fc2.bias")


def _scatter_add_rows(dst, idx, rows):
    """dst[idx[i]] += rows[i] (np.add.at semantics; fp32 sums in the order of a stable sort by index).  np.add.at walks the
    1.1 M level-0 references of a C2 batch one row at a time; a stable sort + reduceat does the same sums 5x faster."""
    if len(idx) < 4096:
        np.add.at(dst, idx, rows)
        return
    order = np.argsort(idx, kind="stable")
    sidx = idx[order]
    starts = np.flatnonzero(np.r_[True, sidx[1:] != sidx[:-1]])
    dst[sidx[starts]] += np.add.reduceat(rows[order], starts, axis=0)


class OracleTGN:
    """State + step semantics of model/tgn.py (graph_attention + identity message + last aggregator + GRU)."""

    def __init__(self, neighbor_finder, node_features, edge_features, params, n_layers, n_heads, use_memory=True):
        self.neighbor_finder = neighbor_finder
        self.node_features = np.asarray(node_features).astype(f32)                 # tgn.py:35
        self.edge_features = normalise_edge_features(edge_features)                 # tgn.py:38-41
        self.P = {k: np.asarray(v, f32) for k, v in params.items()}
        self.n_layers, self.n_heads, self.use_memory = n_layers, n_heads, use_memory
        self.n_nodes, self.D = self.node_features.shape
        # train-mode attention dropout, injected: {layer l (1-based): f32 [n_l, H, K]} multipliers for the n_l instances of
        # layer l in LEVEL ORDER (S_L = roots, S_{l-1} = [S_l ; neighbours(S_l) flattened]: instance i of S_l keeps index i
        # in S_{l-1}, its j-th neighbour sits at |S_l| + i K + j), or None = dropout 0 / eval mode
        self.dropout_masks = None
        self.init_memory()

    # -- modules/memory.py:23-33
    def init_memory(self):
        self.memory = np.zeros((self.n_nodes, self.D), f32)
        self.last_update = np.zeros(self.n_nodes, f32)
        self.messages = defaultdict(list)

    def _w(self):
        return self.P["time_encoder.w.weight"].reshape(-1), self.P["time_encoder.w.bias"]

    def _gru(self):
        pre = "memory_updater.memory_updater."
        return self.P[pre + "weight_ih"], self.P[pre + "weight_hh"], self.P[pre + "bias_ih"], self.P[pre + "bias_hh"]

    # -- message_aggregator.py:38-55
    def _aggregate_last(self, node_ids):
        ids, msgs, ts = [], [], []
        for nid in np.unique(node_ids):
            lst = self.messages.get(int(nid), [])
            if len(lst) > 0:
                ids.append(int(nid)); msgs.append(lst[-1][0]); ts.append(lst[-1][1])
        if ids:
            return np.array(ids, np.int64), np.stack(msgs).astype(f32), np.array(ts, f32)
        return np.zeros(0, np.int64), np.zeros((0, 0), f32), np.zeros(0, f32)

    # -- memory_updater.py:35-53
    def _get_updated_memory(self):
        ids, msgs, ts = self._aggregate_last(np.arange(self.n_nodes))
        mem, lu = self.memory.copy(), self.last_update.copy()
        cache = None
        if len(ids) > 0:
            assert (self.last_update[ids] <= ts).all(), "Trying to update memory to time in the past"
            hn, cache = gru_cell(msgs, mem[ids], *self._gru())
            mem[ids] = hn
            lu[ids] = ts
        return mem, lu, (ids, cache)

    # -- memory_updater.py:18-33
    def _update_memory(self, positives):
        ids, msgs, ts = self._aggregate_last(positives)
        if len(ids) == 0:
            return
        assert (self.last_update[ids] <= ts).all(), "Trying to update memory to time in the past"
        h = self.memory[ids]
        self.last_update[ids] = ts
        self.memory[ids], _ = gru_cell(msgs, h, *self._gru())

    # -- tgn.py:357-378
    def _get_raw_messages(self, src, dst, edge_times, edge_idxs):
        et = np.asarray(edge_times).astype(f32)                                    # :359
        ef = self.edge_features[edge_idxs]
        delta = et - self.last_update[src]                                          # :367 fp32
        enc = time_encode(delta, *self._w())
        msg = np.concatenate([self.memory[src], self.memory[dst], ef, enc], 1).astype(f32)   # :371
        for i in range(len(src)):
            self.messages[int(src[i])].append((msg[i], et[i]))                      # :375-376 (store_raw_messages extends)

    # -- embedding_module.py:76-175 (recursive)
    def _embed(self, memory, nodes, ts, l, K, draws, base=0, n_l=None):
        """``base`` / ``n_l``: position of nodes[0] in the level list S_l and |S_l| (only used to address injected dropout
        masks: the recursion itself is the reference's)."""
        nodes = np.asarray(nodes, np.int64)
        if n_l is None:
            n_l = len(nodes)
        if l == 0:
            feat = self.node_features[nodes]
            if self.use_memory:
                feat = memory[nodes] + feat                                         # :98
            return feat.astype(f32), ("leaf", nodes)
        w, b = self._w()
        Kc = K if K > 0 else 1
        n_below = n_l * (1 + Kc)
        x, c_x = self._embed(memory, nodes, ts, l - 1, K, draws, base, n_below)     # :115
        if draws is not None and self.neighbor_finder.uniform:
            nbr, eidx, et = self.neighbor_finder.gather_uniform(nodes, ts, draws.pop(0), K)
        else:
            nbr, eidx, et = self.neighbor_finder.get_temporal_neighbor(nodes, ts, K)   # :125
        deltas = (ts[:, None] - et).astype(f32)                                     # :133-135 (f64 - f32 -> f32)
        nb, c_nb = self._embed(memory, nbr.flatten(), np.repeat(ts, K), l - 1, K, draws, n_l + base * Kc, n_below)   # :141
        nb = nb.reshape(len(nodes), Kc, -1)
        te = time_encode(deltas, w, b)                                              # :150
        tq = np.broadcast_to(time_encode(np.zeros(1, f32), w, b), (len(nodes), self.D)).astype(f32)   # :92
        ef = self.edge_features[eidx]                                               # :152
        mask = nbr == 0                                                             # :154
        drop = None
        if self.dropout_masks is not None:
            drop = self.dropout_masks[l][base:base + len(nodes)]
            assert drop.shape == (len(nodes), self.n_heads, Kc), (drop.shape, len(nodes), self.n_heads, Kc)
        out, c = attention_forward_mt(layer_params(self.P, l - 1), x, tq, nb, ef, te, mask, self.n_heads, drop)
        return out, ("layer", l, c_x, c_nb, c, deltas, (nbr, eidx, et))

    def _embed_backward(self, ctx, d_out, grads, d_mem):
        if ctx[0] == "leaf":
            if self.use_memory:
                _scatter_add_rows(d_mem, ctx[1], d_out)
            return
        _, l, c_x, c_nb, c, deltas, _ = ctx
        w, b = self._w()
        g, d_x, d_tq, d_nb, d_te = attention_backward_mt(layer_params(self.P, l - 1), c, d_out, self.n_heads, self.D)
        pre = "embedding_module.attention_models.%d." % (l - 1)
        for k, v in g.items():
            grads[pre + _LAYER_KEYS[k]] += v
        gw, gb = time_encode_backward(deltas, w, b, d_te)
        gw0, gb0 = time_encode_backward(np.zeros(d_tq.shape[0], f32), w, b, d_tq)
        grads["time_encoder.w.weight"] += (gw + gw0).reshape(-1, 1)
        grads["time_encoder.w.bias"] += gb + gb0
        self._embed_backward(c_x, d_x, grads, d_mem)
        self._embed_backward(c_nb, d_nb.reshape(-1, self.D), grads, d_mem)

    # -- tgn.py:219-327 / 102-217
    def compute_temporal_embeddings(self, src, dst, neg, edge_times, edge_idxs, n_neighbors=20, draws=None):
        B = len(src)
        size = len(neg) // B
        nodes = np.concatenate([src, dst, neg])
        ts = np.concatenate([edge_times, edge_times, np.repeat(edge_times, size)]).astype(np.float64)
        emb = self._step(nodes, ts, src, dst, edge_times, edge_idxs, n_neighbors, draws)
        return emb[:B], emb[B:2 * B], emb[2 * B:]

    def compute_temporal_embeddings_p(self, src, dst, p_pos, p_neg, edge_times, edge_idxs, n_neighbors=20, draws=None):
        B = len(src)
        npos, nneg = len(p_pos) // B, len(p_neg) // B
        nodes = np.concatenate([src, dst, p_pos, p_neg])
        # tgn.py:124 - p_pos shares the un-repeated edge_times (NUM_POS_TRAIN == 1 in the reference)
        ts = np.concatenate([edge_times, edge_times, np.repeat(edge_times, npos), np.repeat(edge_times, nneg)]).astype(np.float64)
        emb = self._step(nodes, ts, src, dst, edge_times, edge_idxs, n_neighbors, draws)
        return emb[:B], emb[B:2 * B], emb[2 * B:(2 + npos) * B], emb[(2 + npos) * B:]

    def _step(self, nodes, ts, src, dst, edge_times, edge_idxs, K, draws):
        memory, self._gru_ctx = None, None
        if self.use_memory:
            memory, _, self._gru_ctx = self._get_updated_memory()                   # tgn.py:251
        emb, self._ctx = self._embed(memory, nodes, ts, self.n_layers, K, draws)    # tgn.py:275
        if self.use_memory:
            positives = np.concatenate([src, dst])
            self._update_memory(positives)                                          # tgn.py:295
            assert np.allclose(memory[positives], self.memory[positives], atol=1e-5)    # tgn.py:298
            for nid in positives:                                                   # tgn.py:302
                self.messages[int(nid)] = []
            self._get_raw_messages(src, dst, edge_times, edge_idxs)                 # tgn.py:304-317
            self._get_raw_messages(dst, src, edge_times, edge_idxs)
        return emb

    def backward(self, d_emb):
        """d_emb [R,D] for the concatenated roots of the last step -> dict of parameter grads."""
        grads = {k: np.zeros_like(v) for k, v in self.P.items()}
        d_mem = np.zeros((self.n_nodes, self.D), f32)
        self._embed_backward(self._ctx, np.asarray(d_emb, f32), grads, d_mem)
        if self.use_memory and self._gru_ctx is not None and self._gru_ctx[1] is not None:
            ids, cache = self._gru_ctx
            W_ih, W_hh, _, _ = self._gru()
            g = gru_cell_backward(cache, d_mem[ids], W_ih, W_hh)
            for k, v in g.items():
                grads["memory_updater.memory_updater." + k] += v.astype(f32)
        return grads

    # dense view of the pending-message table (last message per node) for comparisons with the device layout
    def pending_table(self):
        M = 3 * self.D + self.edge_features.shape[1]
        tab = np.zeros((self.n_nodes, M), f32); t = np.zeros(self.n_nodes, f32); has = np.zeros(self.n_nodes, bool)
        for nid, lst in self.messages.items():
            if len(lst) > 0:
                tab[nid], t[nid], has[nid] = lst[-1][0], lst[-1][1], True
        return tab, t, has


def init_params(D, Ef, n_layers, seed=0, use_memory=True):
    """Random parameters with the reference's shapes (App. B); time-encoder init per time_encoding.py:13-15."""
    rs = np.random.RandomState(seed)
    E, C, M = 2 * D, 2 * D + Ef, 3 * D + Ef
    P = {"time_encoder.w.weight": (1 / 10 ** np.linspace(0, 9, D)).astype(f32).reshape(D, 1),
         "time_encoder.w.bias": (rs.randn(D) * 0.1).astype(f32)}
    def lin(o, i):
        return (rs.randn(o, i) * np.sqrt(1.0 / i)).astype(f32)
    if use_memory:
        pre = "memory_updater.memory_updater."
        P[pre + "weight_ih"] = lin(3 * D, M); P[pre + "weight_hh"] = lin(3 * D, D)
        P[pre + "bias_ih"] = (rs.randn(3 * D) * 0.1).astype(f32); P[pre + "bias_hh"] = (rs.randn(3 * D) * 0.1).astype(f32)
    for l in range(n_layers):
        pre = "embedding_module.attention_models.%d." % l
        P[pre + "multi_head_target.q_proj_weight"] = lin(E, E)
        P[pre + "multi_head_target.k_proj_weight"] = lin(E, C)
        P[pre + "multi_head_target.v_proj_weight"] = lin(E, C)
        P[pre + "multi_head_target.in_proj_bias"] = (rs.randn(3 * E) * 0.1).astype(f32)
        P[pre + "multi_head_target.out_proj.weight"] = lin(E, E)
        P[pre + "multi_head_target.out_proj.bias"] = (rs.randn(E) * 0.1).astype(f32)
        P[pre + "merger.fc1.weight"] = lin(D, E + D); P[pre + "merger.fc1.bias"] = (rs.randn(D) * 0.1).astype(f32)
        P[pre + "merger.fc2.weight"] = lin(D, D); P[pre + "merger.fc2.bias"] = (rs.randn(D) * 0.1).astype(f32)
    return P
