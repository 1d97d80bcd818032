#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Imports youngandbin/PfoTGNRec from /root/reference (read-only; Python, CPU) on
the synthetic graphs of ``pfotgnrec_amd.synthetic`` and stores inputs plus the
reference's outputs.  The fixtures are data only - no reference source text is
stored.  The inline MV block of main.py (not importable: wandb/CUDA/data files)
is executed in place from the reference tree, as SURVEY.md App. E describes.

Usage:  python tools/make_golden.py [g1 g2 g3 g4 g5 g6 g7 g8]      (default: all)
Library versions used are recorded in each fixture (``versions``).
"""
import os
import sys
import copy
import types
import textwrap
from collections import defaultdict

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import scipy
import scipy.stats
import torch

from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph, split_train  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
VERSIONS = "torch %s numpy %s scipy %s" % (torch.__version__, np.__version__, scipy.__version__)


def ref_modules():
    from utils.utils import get_neighbor_finder, NeighborFinder, RandEdgeSampler, MergeLayer
    from utils.data import Data
    from model.tgn import TGN
    from model.time_encoding import TimeEncode
    from model.temporal_attention import TemporalAttentionLayer
    return types.SimpleNamespace(**locals())


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, versions=np.array(VERSIONS), **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def adversarial_graph():
    """Duplicate timestamps, zero-degree nodes, K > degree, node 0 queried."""
    rs = np.random.RandomState(7)
    U, I, E = 12, 6, 80
    src = rs.randint(1, U - 2, size=E)            # users U-2..U never interact (zero degree)
    dst = rs.randint(U + 1, U + I + 1, size=E)
    ts = np.sort(rs.randint(0, 25, size=E)).astype(np.float64)    # heavy timestamp ties
    eidx = np.arange(1, E + 1)
    return src, dst, ts, eidx, U + I


# ------------------------------------------------------------------ G1: neighbour sampler
def g1():
    R = ref_modules()
    out = {}
    # (a) C1-like graph, most-recent mode
    cfg = SyntheticConfig("g1", 300, 40, 4000, 8, 1, 10, 2)
    g = make_graph(cfg, with_prices=False, with_portfolios=False)
    d = g.data
    nf = R.get_neighbor_finder(R.Data(d.sources, d.destinations, d.timestamps, d.edge_idxs, d.labels, None), uniform=False)
    rs = np.random.RandomState(3)
    q_nodes = np.concatenate([d.sources[2000:2256], d.destinations[2000:2256], rs.randint(0, g.n_nodes, 256)])
    q_ts = np.concatenate([d.timestamps[2000:2256], d.timestamps[2000:2256], rs.randint(0, 1 << 24, 256).astype(np.float64)])
    for K in (10, 3, 0):
        nb, ei, et = nf.get_temporal_neighbor(q_nodes, q_ts, K)
        out["a_K%d_nbr" % K], out["a_K%d_eidx" % K], out["a_K%d_et" % K] = nb, ei, et
    out.update(a_src=d.sources, a_dst=d.destinations, a_ts=d.timestamps, a_eidx=d.edge_idxs, a_q_nodes=q_nodes, a_q_ts=q_ts)

    # (b) adversarial graph, most-recent + uniform (draws logged)
    src, dst, ts, eidx, max_node = adversarial_graph()
    D = R.Data(src, dst, ts, eidx, np.zeros(len(src)), None)
    nf = R.get_neighbor_finder(D, uniform=False)
    q_nodes = np.concatenate([np.arange(0, max_node + 1), src[40:], dst[40:]])
    q_ts = np.concatenate([np.full(max_node + 1, 13.0), ts[40:], ts[40:]])
    for K in (4, 20):
        nb, ei, et = nf.get_temporal_neighbor(q_nodes, q_ts, K)
        out["b_K%d_nbr" % K], out["b_K%d_eidx" % K], out["b_K%d_et" % K] = nb, ei, et
    nfu = R.get_neighbor_finder(D, uniform=True)
    log = []
    orig = np.random.randint

    def rec(lo, hi, n):
        r = orig(lo, hi, n); log.append((hi, r.copy())); return r
    np.random.seed(11)
    np.random.randint = rec
    try:
        nb, ei, et = nfu.get_temporal_neighbor(q_nodes, q_ts, 5)
    finally:
        np.random.randint = orig
    # dense draws: rows without history get -1
    draws = np.full((len(q_nodes), 5), -1, np.int64)
    it = iter(log)
    for i, (n_, t_) in enumerate(zip(q_nodes, q_ts)):
        if len(nf.find_before(n_, t_)[0]) > 0:
            hi, r = next(it); draws[i] = r
    out.update(b_src=src, b_dst=dst, b_ts=ts, b_eidx=eidx, b_q_nodes=q_nodes, b_q_ts=q_ts,
               b_uni_nbr=nb, b_uni_eidx=ei, b_uni_et=et, b_uni_draws=draws, b_uni_seed=np.array(11))
    save("g1_sampler", **out)


# ------------------------------------------------------------------ G2: candidate draw
def g2():
    R = ref_modules()
    cfg = SyntheticConfig("g2", 200, 30, 3000, 8, 1, 10, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    sl = slice(1000, 1064)
    out = dict(src=d.sources[sl], dst_all=d.destinations[:2400], port_idx=g.portfolio_idx[sl], port_len=g.portfolio_len[sl],
               upper_u=np.array(g.upper_u), n_items=np.array(cfg.n_items))
    for size, seed in ((3, None), (20, None), (30, 2024)):      # 30 > available when a portfolio is non-empty -> replace=True
        np.random.seed(5)
        s = R.RandEdgeSampler(d.sources[sl], d.destinations[:2400], d.portfolios[sl], g.upper_u, g.map_item_id, seed=seed)
        out["neg_size%d" % size] = s.sample(size)
        out["seed_size%d" % size] = np.array(-1 if seed is None else seed)
    save("g2_candidates", **out)


# ------------------------------------------------------------------ G3: MV selection (main.py:192-304 executed in place)
class _Rec:
    def __init__(self, mod, hooks):
        self._m, self._h = mod, hooks

    def __getattr__(self, k):
        return self._h.get(k, getattr(self._m, k))


def run_mv_block(g, train_dst, sl, lam, gamma=2.0, num_neg=20, p_pos=1, p_neg=3, seed=9):
    R = ref_modules()
    d = g.data
    src_lines = open(os.path.join(REF, "main.py")).read().split("\n")[191:304]
    code = textwrap.dedent("\n".join(src_lines))
    rank_log, argsort_log, neg_log = [], [], []

    def rankdata(x):
        r = scipy.stats.rankdata(x); rank_log.append((np.array(x, np.float64), r)); return r

    def argsort(x, *a, **k):
        r = np.argsort(x, *a, **k); argsort_log.append((np.array(x, np.float64), r)); return r

    class RecSampler(R.RandEdgeSampler):
        def sample(self, size):
            r = super().sample(size); neg_log.append(r.copy()); return r

    ts_b = d.timestamps[sl]
    time_feature, day_idx = {}, np.zeros(len(ts_b), np.int64)
    for i, ts in enumerate(ts_b):
        key = str(ts)[:8]
        if key not in time_feature:
            day = int(g.day_of(ts))
            time_feature[key] = {"_day": day, **{c: g.prices[day, j] for j, c in enumerate(g.codes)}}
        day_idx[i] = time_feature[key]["_day"]
    ns = dict(np=_Rec(np, {"argsort": argsort}), stats=_Rec(scipy.stats, {"rankdata": rankdata}), RandEdgeSampler=RecSampler,
              args=types.SimpleNamespace(num_negatives=num_neg, p_pos_num=p_pos, p_neg_num=p_neg, gamma=gamma, lambda_mv=lam),
              train_data=types.SimpleNamespace(destinations=train_dst), upper_u=g.upper_u, map_item_id=g.map_item_id,
              time_feature=time_feature, sources_batch=d.sources[sl], destinations_batch=d.destinations[sl].copy(),
              portfolios_batch=d.portfolios[sl], timestamps_batch=ts_b)
    np.random.seed(seed)
    exec(compile(code, "<reference main.py:192-304>", "exec"), ns)
    B = len(ts_b)
    # per interaction: rankdata(y_mv), rankdata(tgn), argsort(y_mv), argsort(new_rank)
    y_mv = np.stack([rank_log[2 * b][0] for b in range(B)])
    invest_rank = np.stack([rank_log[2 * b][1] for b in range(B)])
    new_rank = np.stack([argsort_log[2 * b + 1][0] for b in range(B)])
    order = np.stack([argsort_log[2 * b + 1][1][::-1] for b in range(B)])
    return dict(neg=neg_log[0], y_mv=y_mv, invest_rank=invest_rank, new_rank=new_rank, order=order,
                p_pos=np.asarray(ns["p_pos_batch"]), p_neg=np.asarray(ns["p_neg_batch"]), day_idx=day_idx,
                dst_after=np.asarray(ns["destinations_batch"]))


def g3():
    cfg = SyntheticConfig("g3", 200, 40, 3000, 8, 1, 10, 2, n_days=8)
    g = make_graph(cfg)
    d = g.data
    sl = slice(1500, 1564)
    out = dict(src=d.sources[sl], dst=d.destinations[sl], ts=d.timestamps[sl], dst_all=d.destinations[:2400],
               port_idx=g.portfolio_idx[sl], port_len=g.portfolio_len[sl], prices=g.prices, upper_u=np.array(g.upper_u),
               gamma=np.array(2.0))
    for lam in (0.5, 0.1):
        r = run_mv_block(g, d.destinations[:2400], sl, lam)
        for k, v in r.items():
            out["lam%02d_%s" % (int(lam * 10), k)] = v
    out.update(_g3_layout_arrays(g, sl))
    save("g3_mv", **out)


def _g3_layout_arrays(g, sl):
    """The file-format side of the g3 inputs, as data: the ``time_feature`` day key of every interaction
    (``str(ts)[:8]``, main.py:212 - the keys run_mv_block builds its dict with), the stock codes in item order (the pickle's
    inner keys / ``map_item_id``) and the batch's portfolios as the code lists ``ml_transaction.json`` holds ('' = empty)."""
    d = g.data
    keys = np.array([str(ts)[:8] for ts in d.timestamps[sl]])
    W = max(len(p) for p in d.portfolios[sl])
    ports = np.array([list(p) + [""] * (W - len(p)) for p in d.portfolios[sl]])
    return dict(day_keys=keys, codes=np.array(g.codes), port_codes=ports)


def g3_layout():
    """Adds the layout arrays above to the committed g3 fixture WITHOUT re-running the reference (every other array is kept
    byte for byte): they are inputs derived from the synthetic graph, not reference outputs."""
    cfg = SyntheticConfig("g3", 200, 40, 3000, 8, 1, 10, 2, n_days=8)
    g = make_graph(cfg)
    path = os.path.join(OUT, "g3_mv.npz")
    old = dict(np.load(path, allow_pickle=False))
    assert np.array_equal(old["ts"], g.data.timestamps[1500:1564]) and np.array_equal(old["prices"], g.prices)
    old.update(_g3_layout_arrays(g, slice(1500, 1564)))
    np.savez_compressed(path, **old)
    print("augmented %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ G4: modules (fwd + autograd grads)
def g4():
    R = ref_modules()
    torch.manual_seed(0)
    rs = np.random.RandomState(0)
    D, Ef, K, N, H = 12, 4, 6, 40, 2
    out = dict(D=np.array(D), Ef=np.array(Ef), K=np.array(K), H=np.array(H))
    # time encoder with a non-zero bias, |t*w| up to 1e7 (fp32 FMA sensitivity, SURVEY §7 hard part 1)
    te = R.TimeEncode(D)
    with torch.no_grad():
        te.w.bias.copy_(torch.randn(D) * 0.5)
    t = torch.from_numpy(np.concatenate([rs.randint(0, 1 << 24, 90), [0, 1, 16777215]]).astype(np.float32)).reshape(31, 3)
    y = te(t)
    gy = torch.from_numpy(rs.randn(*y.shape).astype(np.float32))
    y.backward(gy)
    out.update(te_t=t.numpy(), te_w=te.w.weight.detach().numpy(), te_b=te.w.bias.detach().numpy(), te_y=y.detach().numpy(),
               te_gy=gy.numpy(), te_gw=te.w.weight.grad.numpy(), te_gb=te.w.bias.grad.numpy())
    # GRU cell
    M = 3 * D + Ef
    gru = torch.nn.GRUCell(M, D)
    x = torch.randn(N, M); h = torch.randn(N, D)
    hn = gru(x, h)
    ghn = torch.randn(N, D)
    hn.backward(ghn)
    out.update(gru_x=x.numpy(), gru_h=h.numpy(), gru_hn=hn.detach().numpy(), gru_ghn=ghn.numpy(),
               **{"gru_" + k: v.detach().numpy() for k, v in gru.named_parameters()},
               **{"gru_g_" + k: v.grad.numpy() for k, v in gru.named_parameters()})
    # temporal attention layer: all-padding rows, partial masks
    att = R.TemporalAttentionLayer(D, D, Ef, D, output_dimension=D, n_head=H, dropout=0.0)
    xs = torch.randn(N, D, requires_grad=True); tq = torch.randn(N, 1, D, requires_grad=True)
    nb = torch.randn(N, K, D, requires_grad=True); tn = torch.randn(N, K, D, requires_grad=True); ef = torch.randn(N, K, Ef)
    mask = torch.zeros(N, K, dtype=torch.bool)
    mask[:8] = True                         # no valid neighbour at all
    for i in range(8, 24):
        mask[i, :rs.randint(1, K)] = True   # left padding
    o, wts = att(xs, tq, nb, tn, ef, mask.clone())
    go = torch.randn(N, D)
    o.backward(go)
    out.update(att_x=xs.detach().numpy(), att_tq=tq.detach().numpy()[:, 0], att_nb=nb.detach().numpy(), att_tn=tn.detach().numpy(),
               att_ef=ef.numpy(), att_mask=mask.numpy(), att_out=o.detach().numpy(), att_go=go.numpy(), att_w=wts.detach().numpy(),
               att_gx=xs.grad.numpy(), att_gtq=tq.grad.numpy()[:, 0], att_gnb=nb.grad.numpy(), att_gtn=tn.grad.numpy(),
               **{"att_p_" + k: v.detach().numpy() for k, v in att.named_parameters()},
               **{"att_g_" + k: v.grad.numpy() for k, v in att.named_parameters()})
    save("g4_modules", **out)


# ------------------------------------------------------------------ G5: full training steps with injected state
def dense_messages(tgn, n_nodes, M):
    tab = np.zeros((n_nodes, M), np.float32); t = np.zeros(n_nodes, np.float32); cnt = np.zeros(n_nodes, np.int32)
    for nid, lst in tgn.memory.messages.items():
        cnt[nid] = len(lst)
        if lst:
            tab[nid] = lst[-1][0].detach().numpy(); t[nid] = float(lst[-1][1])
    return tab, t, cnt


def g5():
    # Note: every recorded step stores its own inputs (state_dict, memory, pending messages, batch, draws) next to its
    # outputs, so a fixture is self-contained.  The 5-step trajectory itself is not bit-reproducible across runs of this
    # script for the 2-layer memory case (multi-threaded CPU reductions in the reference's torch ops); regenerating
    # replaces that fixture with an equally valid one.
    R = ref_modules()
    for tag, L, use_mem, uniform, H, path in (("L1_mem", 1, True, False, 2, "base"), ("L2_mem", 2, True, False, 2, "base"),
                                               ("L2_nomem_uniform", 2, False, True, 4, "base"), ("L1_mem_p", 1, True, False, 2, "p")):
        torch.manual_seed(1); np.random.seed(1)
        cfg = SyntheticConfig("g5", 120, 20, 1500, 16, L, 5, H)
        g = make_graph(cfg, with_prices=False)
        d = g.data
        D, Ef, K, B, n = cfg.dim, cfg.edge_dim, cfg.n_neighbors, 24, g.n_nodes
        M = 3 * D + Ef
        rdata = R.Data(d.sources, d.destinations, d.timestamps, d.edge_idxs, d.labels, d.portfolios)
        nf = R.get_neighbor_finder(rdata, uniform=uniform)
        tgn = R.TGN(neighbor_finder=nf, node_features=g.node_features, edge_features=g.edge_features.copy(), device=torch.device("cpu"),
                    n_layers=L, n_heads=H, dropout=0.0, use_memory=use_mem, message_dimension=100, memory_dimension=D,
                    memory_update_at_start=True, embedding_module_type="graph_attention", message_function="identity",
                    aggregator_type="last", memory_updater_type="gru", n_neighbors=K)
        with torch.no_grad():
            tgn.time_encoder.w.bias.copy_(torch.randn(D) * 0.3)
        opt = torch.optim.Adam(tgn.parameters(), lr=1e-3)
        rs = np.random.RandomState(2)
        out = dict(L=np.array(L), H=np.array(H), K=np.array(K), use_memory=np.array(use_mem), uniform=np.array(uniform),
                   path=np.array(path), src_all=d.sources, dst_all=d.destinations, ts_all=d.timestamps, eidx_all=d.edge_idxs,
                   node_features=g.node_features, edge_features=g.edge_features, lr=np.array(1e-3))
        n_steps, first = 5, 700
        for step in range(n_steps):
            s = first + step * B
            sb, db, tb, eb = d.sources[s:s + B], d.destinations[s:s + B], d.timestamps[s:s + B], d.edge_idxs[s:s + B]
            n_neg = 3
            neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=(B, n_neg))
            ppos = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B)
            record = step >= 2
            pre = "s%d_" % step
            if record:
                sd = {k: v.detach().numpy().copy() for k, v in tgn.state_dict().items()
                      if not (k.startswith("memory_updater.memory.") or k.startswith("embedding_module.memory.")
                              or k.startswith("embedding_module.time_encoder."))}
                for k, v in sd.items():
                    out[pre + "sd_" + k] = v
                if use_mem:
                    tab, mt, cnt = dense_messages(tgn, n, M)
                    out.update({pre + "msg_tab": tab, pre + "msg_t": mt, pre + "msg_cnt": cnt})
                # Adam moments BEFORE this step (main.py:123 optimizer): with them injected, the post-step parameters
                # ("after_*") are a function of this step's gradients alone
                names = {id(v): k for k, v in tgn.named_parameters()}
                for pp_, st_ in opt.state.items():
                    k = names[id(pp_)]
                    out[pre + "adam_m_" + k] = st_["exp_avg"].detach().numpy().copy()
                    out[pre + "adam_v_" + k] = st_["exp_avg_sq"].detach().numpy().copy()
                    out[pre + "adam_t_" + k] = np.array(int(st_["step"]))      # per tensor: the GRU's lag one step (None grad in step 0)
            # uniform mode: log the draws in call order
            draws, orig = [], np.random.randint
            if uniform:
                cur = []

                def rec(lo, hi, n_):
                    r = orig(lo, hi, n_); cur.append(r.copy()); return r
                orig_gtn = nf.get_temporal_neighbor

                def gtn(nodes, ts, n_neighbors=20):
                    cur.clear()
                    np.random.randint = rec
                    try:
                        res = orig_gtn(nodes, ts, n_neighbors)
                    finally:
                        np.random.randint = orig
                    dense = np.full((len(nodes), n_neighbors), -1, np.int64)
                    it = iter(cur)
                    for i, (a, b) in enumerate(zip(nodes, ts)):
                        if len(nf.find_before(a, b)[0]) > 0:
                            dense[i] = next(it)
                    draws.append(dense)
                    return res
                nf.get_temporal_neighbor = gtn
            tgn.train(); opt.zero_grad()
            if path == "p":
                se, de, pe, ne = tgn.compute_temporal_embeddings_p(sb, db, ppos, neg.flatten(), tb, eb, K)
            else:
                se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg.flatten(), tb, eb, K)
                pe = de
            if uniform:
                nf.get_temporal_neighbor = orig_gtn
            bs = se.shape[0]
            pos_scores = torch.sum(se.view(bs, 1, -1) * pe.view(bs, 1, -1), dim=2)
            neg_scores = torch.matmul(se.view(bs, 1, -1), ne.view(bs, n_neg, -1).transpose(1, 2)).squeeze()
            loss = -torch.mean(torch.log(torch.sigmoid(torch.mean(pos_scores - neg_scores, dim=1))))
            for e_ in (se, de, pe, ne):
                e_.retain_grad()
            loss.backward()
            if record:
                out.update({pre + "src": sb, pre + "dst": db, pre + "ts": tb, pre + "eidx": eb, pre + "neg": neg, pre + "ppos": ppos,
                            pre + "emb_src": se.detach().numpy(), pre + "emb_dst": de.detach().numpy(),
                            pre + "emb_pos": pe.detach().numpy(), pre + "emb_neg": ne.detach().numpy(),
                            pre + "loss": np.array(loss.item(), np.float32),
                            pre + "gemb_src": se.grad.numpy(), pre + "gemb_neg": ne.grad.numpy(),
                            pre + "gemb_pos": (pe.grad.numpy() if pe.grad is not None else np.zeros_like(pe.detach().numpy()))})
                for k, v in tgn.named_parameters():
                    if v.requires_grad:
                        out[pre + "grad_" + k] = (v.grad.numpy().copy() if v.grad is not None else np.zeros(v.shape, np.float32))
                for j, dr in enumerate(draws):
                    out[pre + "draws%d" % j] = dr
            opt.step()
            if use_mem:
                tgn.memory.detach_memory()
            if record:
                for k, v in tgn.named_parameters():
                    if v.requires_grad:
                        out[pre + "after_" + k] = v.detach().numpy().copy()
                if use_mem:
                    tab, mt, cnt = dense_messages(tgn, n, M)
                    out.update({pre + "after_memory": tgn.memory.memory.detach().numpy().copy(),
                                pre + "after_last_update": tgn.memory.last_update.detach().numpy().copy(),
                                pre + "after_msg_tab": tab, pre + "after_msg_t": mt, pre + "after_msg_cnt": cnt})
        out["recorded_steps"] = np.array([2, 3, 4])
        save("g5_step_" + tag, **out)


def g6():
    """Ranking metrics of evaluation.py:114-145 through the reference's own recall_at_k / ndcg_at_k (evaluation.py:11-21):
    score rows [positive | N negatives], ranking = np.argsort(scores)[::-1], positive = index 0.  Rows cover clear
    winners / losers, exact ties between the positive and negatives (the destination itself can be among the negatives:
    utils.py:96 only removes the portfolio), ties among negatives only, and all-equal rows."""
    import evaluation as ev                                   # the reference module (imports cleanly: sklearn, tqdm present)
    rs = np.random.RandomState(11)
    B, N = 96, 25
    scores = rs.randn(B, 1 + N).astype(np.float32)
    kind = np.zeros(B, np.int64)
    for b in range(B):
        k = b % 6
        kind[b] = k
        if k == 1:                                            # the destination appears among the negatives: an exact tie
            scores[b, 1 + rs.randint(N)] = scores[b, 0]
        elif k == 2:                                          # several negatives tie with the positive
            scores[b, 1 + rs.choice(N, 3, replace=False)] = scores[b, 0]
        elif k == 3:                                          # ties among negatives only
            scores[b, 1 + rs.choice(N, 4, replace=False)] = scores[b, 1]
        elif k == 4:                                          # positive on top / at the bottom
            scores[b, 0] = scores[b].max() + 1.0 if b % 12 == 4 else scores[b].min() - 1.0
        elif k == 5 and b % 12 == 5:                          # everything equal
            scores[b, :] = 0.25
    topk = [1, 3, 5]
    pos_rank = np.zeros(B, np.int64)                          # position of the positive in the reference's ranking
    recall = np.zeros((B, 3)); ndcg = np.zeros((B, 3))
    n_greater = (scores[:, 1:] > scores[:, :1]).sum(1)        # the positive's rank lies in [n_greater, n_greater + n_equal]
    n_equal = (scores[:, 1:] == scores[:, :1]).sum(1)
    for b in range(B):
        ranking = np.argsort(scores[b])[::-1]                 # evaluation.py:122
        pos_rank[b] = int(np.where(ranking == 0)[0][0])
        recall[b] = [ev.recall_at_k(ranking, [0], k) for k in topk]   # evaluation.py:127
        ndcg[b] = [ev.ndcg_at_k(ranking, [0], k) for k in topk]       # evaluation.py:128
        assert n_greater[b] <= pos_rank[b] <= n_greater[b] + n_equal[b]
    save("g6_eval_metrics", scores=scores, kind=kind, pos_rank=pos_rank, recall=recall, ndcg=ndcg, n_greater=n_greater,
         n_equal=n_equal, topk=np.array(topk))


def g7():
    """Price ingest the way main.py consumes it: ``time_feature[str(ts)[:8]][code]`` -> 30 prices (main.py:88-89, 212-227),
    log-returns np.log(p[1:] / p[:-1]); the fixture holds a small pickled-dict equivalent as arrays (day keys, codes,
    prices) plus the per-interaction features the reference's expressions produce for them."""
    rs = np.random.RandomState(13)
    days = ["20240102", "20240103", "20240105"]
    codes = ["005930", "000660", "035420", "051910"]
    prices = 100.0 * np.exp(np.cumsum(rs.randn(len(days), len(codes), 30) * 0.02, axis=2))
    time_feature = {dkey: {c: prices[i, j] for j, c in enumerate(codes)} for i, dkey in enumerate(days)}
    map_item_id = {c: j for j, c in enumerate(codes)}
    ts_batch = np.array([202401021530, 202401030915, 202401051200, 202401021000], np.int64)
    cand = [["005930", "035420"], ["000660"], ["051910", "005930", "000660"], ["035420"]]
    feats, mus, shapes = [], [], []
    for ts, stocks in zip(ts_batch, cand):
        ts_ = str(ts)[:8]                                                        # main.py:212
        feature_ = np.array([time_feature[ts_][c] for c in stocks])              # main.py:217
        feature = np.log(feature_[:, 1:] / feature_[:, :-1])                     # main.py:218
        feats.append(feature); mus.append(np.mean(feature, axis=1)); shapes.append(len(stocks))
    save("g7_price_ingest", days=np.array(days), codes=np.array(codes), prices=prices, ts_batch=ts_batch,
         cand_idx=np.array([[map_item_id[c] for c in st] + [-1] * (3 - len(st)) for st in cand]),
         cand_len=np.array(shapes), features=np.concatenate(feats), mus=np.concatenate(mus))


# ------------------------------------------------------------------ G8: train-mode attention dropout (the benched setting)
class DropoutLog:
    """Runs the reference with torch.nn.functional.dropout wrapped: the REAL dropout is called and the multiplier it applied
    is read off its result (1/(1-p) where the weight survived, 0 where it was dropped; entries whose input is 0 - masked
    keys - are recorded as kept: they contribute nothing either way).  nn.MultiheadAttention's explicit path
    (need_weights=True, temporal_attention.py:70) calls it on the softmax weights [N*H, 1, K], row n*H + h."""

    def __init__(self):
        import torch.nn.functional as F
        self.F, self.real, self.masks = F, F.dropout, []

    def __enter__(self):
        def rec(input, p=0.5, training=True, inplace=False):
            out = self.real(input, p, training, False)
            if training and p > 0:
                kept = (out != 0) | (input == 0)
                self.masks.append((kept.float() / (1.0 - p)).numpy().copy())
            return out
        self.F.dropout = rec
        return self

    def __exit__(self, *a):
        self.F.dropout = self.real


def g8():
    R = ref_modules()
    out = {}
    # (a) the layer alone, two rates, all-padding rows and partial masks as in g4
    for tag, pdrop in (("p10", 0.1), ("p50", 0.5)):
        torch.manual_seed(3); rs = np.random.RandomState(3)
        D, Ef, K, N, H = 12, 4, 6, 40, 2
        att = R.TemporalAttentionLayer(D, D, Ef, D, output_dimension=D, n_head=H, dropout=pdrop)
        att.train()
        with torch.no_grad():
            att.multi_head_target.in_proj_bias.normal_(0, 0.2); att.multi_head_target.out_proj.bias.normal_(0, 0.2)
        xs = torch.randn(N, D, requires_grad=True); tq = torch.randn(N, 1, D, requires_grad=True)
        nb = torch.randn(N, K, D, requires_grad=True); tn = torch.randn(N, K, D, requires_grad=True); ef = torch.randn(N, K, Ef)
        mask = torch.zeros(N, K, dtype=torch.bool)
        mask[:8] = True
        for i in range(8, 24):
            mask[i, :rs.randint(1, K)] = True
        with DropoutLog() as log:
            o, wts = att(xs, tq, nb, tn, ef, mask.clone())
        assert len(log.masks) == 1 and log.masks[0].shape == (N * H, 1, K)
        go = torch.randn(N, D)
        o.backward(go)
        pre = "att_%s_" % tag
        out.update({pre + "p": np.array(pdrop), pre + "drop": log.masks[0].reshape(N, H, K),
                    pre + "x": xs.detach().numpy(), pre + "tq": tq.detach().numpy()[:, 0], pre + "nb": nb.detach().numpy(),
                    pre + "tn": tn.detach().numpy(), pre + "ef": ef.numpy(), pre + "mask": mask.numpy(),
                    pre + "out": o.detach().numpy(), pre + "go": go.numpy(),
                    pre + "gx": xs.grad.numpy(), pre + "gtq": tq.grad.numpy()[:, 0], pre + "gnb": nb.grad.numpy(), pre + "gtn": tn.grad.numpy(),
                    **{pre + "p_" + k: v.detach().numpy() for k, v in att.named_parameters()},
                    **{pre + "g_" + k: v.grad.numpy() for k, v in att.named_parameters()}})
    out.update(D=np.array(12), Ef=np.array(4), K=np.array(6), H=np.array(2))
    # (b) one full training step of a 2-layer TGN with memory at dropout 0.1 (main.py:33 default --drop_out 0.1), state
    # injected like g5; the masks of the three attention calls in call order (layer 1 on the roots, layer 1 on their
    # neighbours, layer 2 on the roots: embedding_module.py:115,141,159)
    torch.manual_seed(2); np.random.seed(2)
    cfg = SyntheticConfig("g8", 120, 20, 1500, 16, 2, 5, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    D, Ef, K, B, n, L, H = cfg.dim, cfg.edge_dim, cfg.n_neighbors, 24, g.n_nodes, 2, 2
    M = 3 * D + Ef
    rdata = R.Data(d.sources, d.destinations, d.timestamps, d.edge_idxs, d.labels, d.portfolios)
    nf = R.get_neighbor_finder(rdata, uniform=False)
    tgn = R.TGN(neighbor_finder=nf, node_features=g.node_features, edge_features=g.edge_features.copy(), device=torch.device("cpu"),
                n_layers=L, n_heads=H, dropout=0.1, use_memory=True, message_dimension=100, memory_dimension=D,
                memory_update_at_start=True, embedding_module_type="graph_attention", message_function="identity",
                aggregator_type="last", memory_updater_type="gru", n_neighbors=K)
    with torch.no_grad():
        tgn.time_encoder.w.bias.copy_(torch.randn(D) * 0.3)
    opt = torch.optim.Adam(tgn.parameters(), lr=1e-3)
    rs = np.random.RandomState(4)
    out.update(step_L=np.array(L), step_H=np.array(H), step_K=np.array(K), step_p=np.array(0.1),
               src_all=d.sources, dst_all=d.destinations, ts_all=d.timestamps, eidx_all=d.edge_idxs,
               node_features=g.node_features, edge_features=g.edge_features)
    for step in range(3):
        s = 700 + step * B
        sb, db, tb, eb = d.sources[s:s + B], d.destinations[s:s + B], d.timestamps[s:s + B], d.edge_idxs[s:s + B]
        neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=(B, 3))
        record = step == 2
        if record:
            for k, v in tgn.state_dict().items():
                if not (k.startswith("memory_updater.memory.") or k.startswith("embedding_module.memory.")
                        or k.startswith("embedding_module.time_encoder.")):
                    out["s_sd_" + k] = v.detach().numpy().copy()
            tab, mt, cnt = dense_messages(tgn, n, M)
            out.update(s_msg_tab=tab, s_msg_t=mt, s_msg_cnt=cnt)
        tgn.train(); opt.zero_grad()
        with DropoutLog() as log:
            se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg.flatten(), tb, eb, K)
        bs = se.shape[0]
        pos_scores = torch.sum(se.view(bs, 1, -1) * de.view(bs, 1, -1), dim=2)
        neg_scores = torch.matmul(se.view(bs, 1, -1), ne.view(bs, 3, -1).transpose(1, 2)).squeeze()
        loss = -torch.mean(torch.log(torch.sigmoid(torch.mean(pos_scores - neg_scores, dim=1))))
        loss.backward()
        if record:
            Rr = 5 * B
            assert [m.shape[0] for m in log.masks] == [Rr * H, Rr * K * H, Rr * H], [m.shape for m in log.masks]
            out.update(s_src=sb, s_dst=db, s_ts=tb, s_eidx=eb, s_neg=neg,
                       s_drop_l1=np.concatenate([log.masks[0].reshape(Rr, H, K), log.masks[1].reshape(Rr * K, H, K)]),
                       s_drop_l2=log.masks[2].reshape(Rr, H, K),
                       s_emb_src=se.detach().numpy(), s_emb_dst=de.detach().numpy(), s_emb_neg=ne.detach().numpy(),
                       s_loss=np.array(loss.item(), np.float32))
            for k, v in tgn.named_parameters():
                if v.requires_grad:
                    out["s_grad_" + k] = (v.grad.numpy().copy() if v.grad is not None else np.zeros(v.shape, np.float32))
        opt.step()
        tgn.memory.detach_memory()
        if record:
            tab, mt, cnt = dense_messages(tgn, n, M)
            out.update(s_after_memory=tgn.memory.memory.detach().numpy().copy(),
                       s_after_last_update=tgn.memory.last_update.detach().numpy().copy(),
                       s_after_msg_tab=tab, s_after_msg_t=mt, s_after_msg_cnt=cnt)
    save("g8_dropout", **out)



if __name__ == "__main__":
    import warnings
    warnings.filterwarnings("ignore")
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"]
    for w in which:
        globals()[w]()
