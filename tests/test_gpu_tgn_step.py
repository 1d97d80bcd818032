"""GPU parity of the whole TGN step through the drop-in surface: against the golden fixtures captured from the
reference (state re-injected at every step, SURVEY §7 hard part 5) and against the oracle on larger shapes."""
import numpy as np
import pytest
import torch

from conftest import load_golden, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]

import pfotgnrec_amd as P
from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
from oracle import tgn_oracle as T
from oracle.neighbor_finder import OracleNeighborFinder, build_adjacency

DEV = "cuda:0"
RTOL_EMB = 1e-4      # BASELINE.json north_star: embeddings within 1e-4 relative
RTOL_GRAD = 5e-4     # parameter gradients (atomic accumulation order + folded projections)
# time-encoder gradients are sums of terms scaled by dt ~ 1e7 that cancel to a small remainder: relative to
# max|grad| both the reference's fp32 autograd sum and any re-association of it carry ~1e-3 evaluation noise
RTOL_GRAD_TIME = 3e-3
# oracle comparisons on random parameters: a fc1 pre-activation within rounding distance of 0 flips relu' between
# the two implementations and perturbs every upstream gradient by ~1e-3 (observed once in the D=64 case; all other
# tensors / cases agree to ~1e-5, tools/grad_error_survey.py).  Relative L2 is the metric, with room for one kink.
RTOL_GRAD_ORACLE_L2 = 5e-3   # relative L2 vs the oracle: float-rounding-level differences in the forward flip individual ReLU
                             # units (kinks); two builds of this library that agree with each other to 6e-7 sit at 2.0e-3 and 3.6e-3
                             # against the oracle on the H=4 / uniform configuration.  The reference goldens pin 5e-4 (max norm).


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


def inject(tgn, g, pre):
    sd = tgn.state_dict()
    with torch.no_grad():
        for k in g.files:
            if k.startswith(pre + "sd_"):
                name = k[len(pre + "sd_"):]
                if name in sd:
                    sd[name].copy_(torch.from_numpy(g[k]))
        if tgn.use_memory:
            m = tgn.memory
            m.msg_table.copy_(torch.from_numpy(g[pre + "msg_tab"]))
            m.msg_time.copy_(torch.from_numpy(g[pre + "msg_t"]))
            m.has_msg.copy_(torch.from_numpy((g[pre + "msg_cnt"] > 0).astype(np.uint8)))


@pytest.mark.parametrize("tag", ["L1_mem", "L2_mem", "L2_nomem_uniform", "L1_mem_p"])
def test_step_against_reference_golden(tag):
    g = load_golden("g5_step_" + tag)
    L, H, K = int(g["L"]), int(g["H"]), int(g["K"])
    use_mem, uniform, path = bool(g["use_memory"]), bool(g["uniform"]), str(g["path"])
    nf = P.NeighborFinder.from_arrays(g["src_all"], g["dst_all"], g["eidx_all"], g["ts_all"], uniform=uniform)
    D = g["node_features"].shape[1]
    tgn = P.TGN(nf, g["node_features"], g["edge_features"], DEV, n_layers=L, n_heads=H, dropout=0.0, use_memory=use_mem,
                memory_dimension=D, message_function="identity", n_neighbors=K)
    opt = P.FusedAdam(tgn, lr=float(g["lr"]))
    for step in g["recorded_steps"]:
        pre = "s%d_" % step
        inject(tgn, g, pre)
        sb, db, tb, eb, neg = g[pre + "src"], g[pre + "dst"], g[pre + "ts"], g[pre + "eidx"], g[pre + "neg"]
        B = len(sb)
        draws = None
        if uniform:   # reference call order: layer-1(roots) = draws0, layer-2(roots) = draws1, layer-1(neighbours) = draws2
            draws = [g[pre + "draws1"], np.concatenate([g[pre + "draws0"], g[pre + "draws2"]])] if L == 2 else [g[pre + "draws0"]]
        tgn.train()
        opt.zero_grad()
        if path == "p":
            se, de, pe, ne = tgn.compute_temporal_embeddings_p(sb, db, g[pre + "ppos"], neg.flatten(), tb, eb, K, draws=draws)
            emb = torch.cat([se, de, pe, ne]); pos_block = 2
        else:
            se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg.flatten(), tb, eb, K, draws=draws)
            pe = de
            emb = torch.cat([se, de, ne]); pos_block = 1
        for got, key in ((se, "emb_src"), (de, "emb_dst"), (pe, "emb_pos"), (ne, "emb_neg")):
            e = relerr(got.detach().cpu().numpy(), g[pre + key])
            assert e < RTOL_EMB, (tag, step, key, e)
        loss = P.bpr_loss(emb, B, 3, pos_block=pos_block)
        assert abs(float(loss) - float(g[pre + "loss"])) < 1e-5 * max(1.0, abs(float(g[pre + "loss"])))
        loss.backward()
        for k in g.files:
            if not k.startswith(pre + "grad_"):
                continue
            name = k[len(pre + "grad_"):]
            if "layer_norm" in name or name.startswith("memory."):
                continue
            ref = g[k]
            got = dict(tgn.named_parameters())[name].grad.cpu().numpy()
            scale = np.abs(ref).max()
            if scale < 1e-7:           # e.g. the key bias: its gradient cancels exactly in the softmax
                assert np.abs(got).max() < 1e-6, name
                continue
            e = relerr(got, ref)
            assert e < (RTOL_GRAD_TIME if name.startswith("time_encoder") else RTOL_GRAD), (tag, step, name, e)
        if use_mem:
            assert relerr(tgn.memory.memory.cpu().numpy(), g[pre + "after_memory"]) < RTOL_EMB
            assert np.array_equal(tgn.memory.last_update.cpu().numpy(), g[pre + "after_last_update"])
            assert np.array_equal(tgn.memory.has_msg.cpu().numpy() > 0, g[pre + "after_msg_cnt"] > 0)
            has = g[pre + "after_msg_cnt"] > 0
            assert relerr(tgn.memory.msg_table.cpu().numpy()[has], g[pre + "after_msg_tab"][has]) < RTOL_EMB
            assert np.array_equal(tgn.memory.msg_time.cpu().numpy()[has], g[pre + "after_msg_t"][has])
        # optimizer step (Adam, main.py:123,389) with the reference optimizer's moments injected: post-step parameters
        # against the reference's `after_*`.  Adam divides by sqrt(v): where an element's gradient history is tiny the
        # update amplifies the (tolerated) gradient difference, so the bound per element is derived from it:
        #   |dp| <= lr * (1 - b1) / bc1 * |g - g_ref| / (sqrt(v_ref' / bc2) + eps)      (+ fp32 rounding of p)
        names = dict(tgn.named_parameters())
        opt._m = torch.zeros_like(tgn.flat_parameters)
        opt._v = torch.zeros_like(tgn.flat_parameters)
        views = {p_: (off, n) for p_, off, n, _ in tgn._views}
        grads_now = {}
        for name, p_ in names.items():
            if p_ not in views or (pre + "adam_m_" + name) not in g.files:
                continue
            off, n = views[p_]
            opt._m[off:off + n] = torch.from_numpy(g[pre + "adam_m_" + name].ravel()).to(DEV)
            opt._v[off:off + n] = torch.from_numpy(g[pre + "adam_v_" + name].ravel()).to(DEV)
            grads_now[name] = p_.grad.detach().cpu().numpy().astype(np.float64)
            opt.set_steps({name: int(g[pre + "adam_t_" + name])})     # torch.optim.Adam counts steps per tensor
        assert len(grads_now) >= 10
        if use_mem:   # the GRU tensors had no gradient in the reference's step 0 (no pending message): one step behind
            assert int(g[pre + "adam_t_memory_updater.memory_updater.weight_ih"]) == int(g[pre + "adam_t_time_encoder.w.weight"]) - 1
        opt.step()
        lr, b1, b2, eps = float(g["lr"]), 0.9, 0.999, 1e-8
        for name, gn in grads_now.items():
            t = int(g[pre + "adam_t_" + name]) + 1
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            ref_after = g[pre + "after_" + name].astype(np.float64)
            got_after = names[name].detach().cpu().numpy().astype(np.float64)
            g_ref = g[pre + "grad_" + name].astype(np.float64)
            v_new = b2 * g[pre + "adam_v_" + name].astype(np.float64) + (1 - b2) * g_ref * g_ref
            bound = lr * (1 - b1) / bc1 * np.abs(gn - g_ref) / (np.sqrt(v_new / bc2) + eps)
            slack = 4e-7 * np.maximum(1.0, np.abs(ref_after)) + 2e-3 * lr          # fp32 rounding of p and of m / sqrt(v)
            assert np.all(np.abs(got_after - ref_after) <= 1.5 * bound + slack), (tag, step, name,
                                                                                 float(np.abs(got_after - ref_after).max()))
            # and the update itself is Adam's: recomputed in fp64 from OUR gradient and the injected moments
            m_new = b1 * g[pre + "adam_m_" + name].astype(np.float64) + (1 - b1) * gn
            v_mine = b2 * g[pre + "adam_v_" + name].astype(np.float64) + (1 - b2) * gn * gn
            want = g[pre + "sd_" + name].astype(np.float64) - lr / bc1 * m_new / (np.sqrt(v_mine / bc2) + eps)
            assert np.abs(got_after - want).max() <= 4e-7 * max(1.0, np.abs(want).max()) + 1e-3 * lr, (tag, step, name)


@pytest.mark.parametrize("D,H,L,K,use_mem,uniform", [(32, 2, 1, 10, True, False), (172, 2, 2, 8, True, False),
                                                      (172, 4, 2, 6, False, True), (64, 1, 2, 5, True, False),
                                                      (24, 4, 3, 3, True, False), (36, 2, 2, 4, True, False),
                                                      (52, 1, 1, 7, False, False),
                                                      (256, 4, 1, 12, True, False),      # widest rows the kernels take (NR = 4)
                                                      (128, 2, 2, 20, True, False),      # K = 20 through the run-merged backward
                                                      (172, 2, 2, 8, True, "scaled")])   # weight blocks binades apart (below)
def test_step_against_oracle(D, H, L, K, use_mem, uniform):
    # "scaled": most-recent sampling, with the weight blocks that meet in ONE contraction moved 2^9-2^10 apart (W_ih against
    # W_hh in the fused GRU, fc1's attention half against its node half and the query weights against them in the two-source
    # launches): the per-row power-of-two scales of the fp16 weight images then differ by ~10 binades between the sources of a
    # launch, and the accumulators are moved from one image's scales to the other's in mid-contraction
    scaled = uniform == "scaled"
    uniform = bool(uniform) and not scaled
    torch.manual_seed(1234 + D + H)
    cfg = SyntheticConfig("t", 300, 25, 5000, D, L, K, H)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    nf = P.get_neighbor_finder(d, uniform=uniform)
    tgn = P.TGN(nf, g.node_features, g.edge_features, DEV, n_layers=L, n_heads=H, dropout=0.0, use_memory=use_mem,
                memory_dimension=D, message_function="identity", n_neighbors=K)
    with torch.no_grad():
        tgn.time_encoder.w.bias.normal_(0, 0.3)
        for att in tgn.embedding_module.attention_models:
            att.multi_head_target.in_proj_bias.normal_(0, 0.1)
            att.multi_head_target.out_proj.bias.normal_(0, 0.1)
        if scaled:
            gru = tgn.memory_updater.memory_updater
            gru.weight_ih.mul_(2.0 ** 3); gru.weight_hh.mul_(2.0 ** -6)
            for att in tgn.embedding_module.attention_models:
                E_ = att.multi_head_target.out_proj.weight.shape[0]
                att.merger.fc1.weight[:, :E_].mul_(2.0 ** 2)
                att.merger.fc1.weight[:, E_:].mul_(2.0 ** -8)
                att.merger.fc2.weight.mul_(2.0 ** -4)                 # (keeps the scores of the loss in range)
                att.multi_head_target.q_proj_weight.mul_(2.0 ** 2)
    opt = P.FusedAdam(tgn, lr=1e-3)
    onf = OracleNeighborFinder(*build_adjacency(d.sources, d.destinations, d.edge_idxs, d.timestamps), uniform=uniform)
    names = [k for k in tgn.state_dict() if "layer_norm" not in k and not k.startswith("memory.")]
    ref = T.OracleTGN(onf, g.node_features, g.edge_features, {k: tgn.state_dict()[k].cpu().numpy() for k in names}, L, H, use_mem)
    rs = np.random.RandomState(5)
    B = 40
    for step in range(4):
        s = 2500 + step * B
        sb, db, tb, eb = d.sources[s:s + B], d.destinations[s:s + B], d.timestamps[s:s + B], d.edge_idxs[s:s + B]
        neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)
        ref.P = {k: tgn.state_dict()[k].detach().cpu().numpy().copy() for k in names}
        draws, odraws = None, None
        if uniform:
            # draw positions on the host so that both sides see the same neighbourhoods
            R = 5 * B
            sizes = [R * (1 + K) ** i for i in range(L)]                    # level L, L-1, ..., 1
            draws = [rs.randint(0, 1 << 30, size=(n, K)).astype(np.int64) for n in sizes]
        tgn.train(); opt.zero_grad()
        if uniform:
            # positions must be < history length: reduce modulo the count on both sides via a shared helper
            draws, odraws = _legal_draws(onf, np.concatenate([sb, db, neg]), np.concatenate([tb, tb, np.repeat(tb, 3)]), K, L, draws)
        se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, K, draws=draws)
        rse, rde, rne = ref.compute_temporal_embeddings(sb, db, neg, tb, eb, K, draws=odraws)
        emb = torch.cat([se, de, ne])
        remb = np.concatenate([rse, rde, rne])
        assert relerr(emb.detach().cpu().numpy(), remb) < RTOL_EMB, (step, relerr(emb.detach().cpu().numpy(), remb))
        rgrads = _masked_bpr_backward(tgn, ref, emb, rse, rde, rne, B, K)
        tol, tol_time = RTOL_GRAD_ORACLE_L2, RTOL_GRAD_TIME
        for name, p in tgn.named_parameters():
            if name not in rgrads:
                continue
            r = rgrads[name].reshape(p.shape)
            if np.abs(r).max() < 1e-7:
                assert p.grad is None or p.grad.abs().max().item() < 1e-6, name
                continue
            got = p.grad.cpu().numpy().astype(np.float64)
            e = np.linalg.norm(got - r) / (np.linalg.norm(r) + 1e-30)
            assert e < (tol_time if name.startswith("time_encoder") else tol), (step, name, e)
        if use_mem:
            assert relerr(tgn.memory.memory.cpu().numpy(), ref.memory) < RTOL_EMB
            assert np.array_equal(tgn.memory.last_update.cpu().numpy(), ref.last_update)
            tab, mt, has = ref.pending_table()
            assert np.array_equal(tgn.memory.has_msg.cpu().numpy() > 0, has)
            assert relerr(tgn.memory.msg_table.cpu().numpy()[has], tab[has]) < RTOL_EMB
            assert np.array_equal(tgn.memory.msg_time.cpu().numpy()[has], mt[has])
            # layer-0 table of the touched nodes = lazily updated memory + node features (embedding_module.py:98)
        opt.step()


KINK_THR = 2e-5              # |fc1 pre-activation| below this on the oracle side: the ReLU decision could differ between the two
                             # implementations (their forwards agree to ~1e-6) - tests/test_gpu_full_size.py uses the same bound


def _near_kink_roots(ctx, R, K, thr=KINK_THR):
    """Roots whose computation tree (embedding_module.py:110-175 recursion, as cached by the oracle) holds a MergeLayer
    fc1 pre-activation within ``thr`` of zero.  Returns bool[R]."""
    bad = np.zeros(R, bool)

    def walk(c, owners):               # owners[i] = root that instance i of this context belongs to
        if c[0] == "leaf":
            return
        _, l, c_x, c_nb, cache, _, _ = c
        near = (np.abs(cache["z1"]) < thr).any(1)
        np.logical_or.at(bad, owners[near], True)
        walk(c_x, owners)
        walk(c_nb, np.repeat(owners, K))
    walk(ctx, np.arange(R))
    return bad


def _masked_bpr_backward(tgn, ref, emb, rse, rde, rne, B, K, n_neg=3):
    """BPR loss on both sides, its embedding gradient compared (2e-5 of its largest entry), then the backward of BOTH sides
    from its own gradient with the rows of near-kink roots zeroed: a ReLU unit whose pre-activation sits within fp32 noise of
    zero may take the other branch here than in the oracle, and one flipped unit moves every gradient below it by ~1/R
    (measured 1-2.5 % at R = 200, tools/probes/time_grad_error.py).  Such a root is left out ON BOTH SIDES instead of
    loosening the bound for the whole step; a step without one (most) is the plain ``loss.backward()``.
    Returns the oracle's parameter gradients; the product's are in ``p.grad``."""
    loss = P.bpr_loss(emb, B, n_neg)
    (d_emb,) = torch.autograd.grad(loss, emb, retain_graph=True)
    rl, cache = T.bpr_loss(rse, rde.reshape(B, 1, -1), rne.reshape(B, n_neg, -1))
    assert abs(float(loss) - float(rl)) < 1e-5
    ds, dp, dn = T.bpr_loss_backward(cache)
    W = np.concatenate([ds, dp.reshape(B, -1), dn.reshape(n_neg * B, -1)]).astype(np.float32)
    # main.py:321-337 backward; both sides start from their OWN embeddings (which agree to ~1e-6 relative)
    assert np.abs(d_emb.cpu().numpy() - W).max() <= 2e-5 * np.abs(W).max() + 1e-9
    R = W.shape[0]
    bad = _near_kink_roots(ref._ctx, R, K)
    assert bad.sum() < R // 2, bad.sum()
    keep = torch.from_numpy((~bad).astype(np.float32)).to(emb.device)[:, None]
    emb.backward(d_emb * keep)
    W[bad] = 0
    return ref.backward(W)


def _legal_draws(onf, roots, ts, K, L, raw):
    """Turns raw random integers into legal per-query positions, level by level, for both call conventions.

    Product order: one tensor per level (L, L-1, ..., 1), level l covering S_l.  Oracle order: the recursion's call
    order.  Levels are expanded with the oracle's own gather so both agree on the frontier.
    """
    nodes, tss = np.asarray(roots, np.int64), np.asarray(ts, np.float64)
    prod = []
    level_nodes = [(nodes, tss)]
    for i in range(L):
        n_, t_ = level_nodes[-1]
        cnt = np.array([len(onf.find_before(int(a), b)[0]) for a, b in zip(n_, t_)])
        dr = np.where(cnt[:, None] > 0, raw[i] % np.maximum(cnt, 1)[:, None], -1)
        prod.append(dr)
        nb, _, _ = onf.gather_uniform(n_, t_, np.maximum(dr, 0), K)
        level_nodes.append((np.concatenate([n_, nb.flatten()]), np.concatenate([t_, np.repeat(t_, K)])))
    # oracle recursion order for L layers: embed(l, S) = embed(l-1, S) ; sample(S) ; embed(l-1, nbrs(S))
    R = len(nodes)

    def rec(l, lo, hi, level):   # rows [lo, hi) of the product's level tensor `level` (0 = roots level)
        out = []
        if l == 0:
            return out
        out += rec(l - 1, lo, hi, level + 1) if level + 1 < L else []
        out.append(prod[level][lo:hi])
        if level + 1 < L:
            n_level = len(level_nodes[level][0])
            out += rec(l - 1, n_level + lo * K, n_level + hi * K, level + 1)
        return out
    return prod, rec(L, 0, R, 0)


def test_eval_mode_no_grad_and_state_progression():
    cfg = SyntheticConfig("t", 200, 20, 3000, 32, 1, 10, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=1, n_heads=2, dropout=0.3,
                use_memory=True, memory_dimension=32, message_function="identity")
    tgn.eval()
    with torch.no_grad():
        a = tgn.compute_temporal_embeddings(d.sources[1000:1032], d.destinations[1000:1032], d.destinations[1000:1096],
                                            d.timestamps[1000:1032], d.edge_idxs[1000:1032], 10)
    assert not a[0].requires_grad and torch.isfinite(a[0]).all()
    assert tgn.memory.has_msg.sum().item() == len(set(d.sources[1000:1032]) | set(d.destinations[1000:1032]))
    bk = tgn.memory.backup_memory()
    tgn.memory.__init_memory__()
    assert tgn.memory.has_msg.sum().item() == 0 and tgn.memory.memory.abs().sum().item() == 0
    tgn.memory.restore_memory(bk)
    assert tgn.memory.has_msg.sum().item() > 0


def test_dropout_training_is_consistent_between_forward_and_backward():
    """finite-difference check through the dropout path: the backward regenerates the forward's Philox mask"""
    torch.manual_seed(77)
    cfg = SyntheticConfig("t", 100, 12, 1500, 16, 1, 6, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=1, n_heads=2, dropout=0.5,
                use_memory=False, memory_dimension=16)
    tgn.train()
    sl = slice(800, 816)
    args = (d.sources[sl], d.destinations[sl], d.destinations[800:848], d.timestamps[sl], d.edge_idxs[sl], 6)

    def run():
        tgn._step = 7                       # same Philox stream each time
        se, de, ne = tgn.compute_temporal_embeddings(*args)
        return torch.cat([se, de, ne])
    emb = run()
    w = torch.randn_like(emb)
    (emb * w).sum().backward()
    p = tgn.embedding_module.attention_models[0].multi_head_target.v_proj_weight
    gnum = p.grad.clone()
    idx = (3, 5)
    eps = 1e-2
    def bump(delta):                      # grad mode stays on for run(): under no_grad dropout is off
        with torch.no_grad():
            p[idx] += delta
    bump(eps)
    up = (run() * w).sum().item()
    bump(-2 * eps)
    dn = (run() * w).sum().item()
    bump(eps)
    fd = (up - dn) / (2 * eps)
    assert abs(fd - gnum[idx].item()) < 5e-2 * max(1.0, abs(fd)), (fd, gnum[idx].item())   # fp32 central difference
    e1, e2 = run(), run()
    assert torch.equal(e1, e2)


def test_evaluation_fast_path_chunked_roots_and_rank_metrics():
    """evaluation.py:88-145: every interaction scores ALL items (R = B*(2+N_ITEMS) roots, forward only).  Roots are
    walked in chunks through the same kernels: identical embeddings and identical memory/message state as one pass;
    ranking metrics on the device equal the host computation."""
    torch.manual_seed(3)
    cfg = SyntheticConfig("ev", 300, 25, 5000, 32, 2, 6, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    B, n_items = 20, cfg.n_items
    sl = slice(3000, 3000 + B)
    neg = np.tile(np.arange(cfg.n_users + 1, cfg.n_users + 1 + n_items), B)            # all items for every interaction
    outs, states = [], []
    for cap in (1 << 30, 96):
        torch.manual_seed(3)
        tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.2,
                    use_memory=True, memory_dimension=32, message_function="identity")
        tgn.eval_chunk_roots = cap
        tgn.eval()
        with torch.no_grad():
            for warm in (2900, 2950):                                                     # populate memory / messages
                w = slice(warm, warm + B)
                tgn.compute_temporal_embeddings(d.sources[w], d.destinations[w], d.destinations[w].repeat(3), d.timestamps[w],
                                                d.edge_idxs[w], 6)
            se, de, ne = tgn.compute_temporal_embeddings(d.sources[sl], d.destinations[sl], neg, d.timestamps[sl], d.edge_idxs[sl], 6)
        outs.append(torch.cat([se, de, ne]))
        states.append((tgn.memory.memory.clone(), tgn.memory.last_update.clone(), tgn.memory.msg_table.clone(), tgn.memory.has_msg.clone()))
    assert outs[0].shape[0] == B * (2 + n_items)
    assert torch.equal(outs[0], outs[1])
    for a, b in zip(states[0], states[1]):
        assert torch.equal(a, b)
    emb = outs[0]
    # the same evaluation batch through the oracle: eval-mode embeddings of all B * (2 + N_ITEMS) roots and the state after
    onf = OracleNeighborFinder(*build_adjacency(d.sources, d.destinations, d.edge_idxs, d.timestamps))
    torch.manual_seed(3)
    tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.2,
                use_memory=True, memory_dimension=32, message_function="identity")
    names = [k for k in tgn.state_dict() if "layer_norm" not in k and not k.startswith("memory.")]
    ref = T.OracleTGN(onf, g.node_features, g.edge_features, {k: tgn.state_dict()[k].cpu().numpy() for k in names}, 2, 2, True)
    for warm in (2900, 2950):
        w = slice(warm, warm + B)
        ref.compute_temporal_embeddings(d.sources[w], d.destinations[w], d.destinations[w].repeat(3), d.timestamps[w], d.edge_idxs[w], 6)
    rse, rde, rne = ref.compute_temporal_embeddings(d.sources[sl], d.destinations[sl], neg, d.timestamps[sl], d.edge_idxs[sl], 6)
    assert relerr(emb.cpu().numpy(), np.concatenate([rse, rde, rne])) < RTOL_EMB
    assert relerr(states[0][0].cpu().numpy(), ref.memory) < RTOL_EMB and np.array_equal(states[0][1].cpu().numpy(), ref.last_update)
    # duplicate (node, time) roots are embedded once: two interactions at one timestamp share their 25 item roots
    tgn.eval(); tgn.eval_dedup = True
    ts_dup = d.timestamps[sl].copy(); ts_dup[1::2] = ts_dup[0::2]
    big = np.tile(neg, 9)                                                # R = 20 * 9 * 25 + 40 >= 4096: the dedup path
    bsrc, bdst, bts, bei = (np.tile(a, 9) for a in (d.sources[sl], d.destinations[sl], ts_dup, d.edge_idxs[sl]))
    res = []
    for flag in (True, False):
        tgn.eval_dedup = flag
        tgn.memory.__init_memory__()
        with torch.no_grad():
            res.append(torch.cat(tgn.compute_temporal_embeddings(bsrc, bdst, big, bts, bei, 6)))
    assert res[0].shape[0] == 180 * 27 and torch.equal(res[0], res[1])
    rank, hits, ndcg = P.rank_metrics(emb, B, n_items)
    e = emb.cpu().numpy().astype(np.float64)
    src, dst, ngs = e[:B], e[B:2 * B], e[2 * B:].reshape(B, n_items, -1)
    pos = (src * dst).sum(1)
    negs = np.einsum("bd,bkd->bk", src, ngs)
    r = rank.cpu().numpy()
    for b in range(B):
        margin = np.abs(negs[b] - pos[b]).min()
        if margin > 1e-5:                                                                 # away from fp32 ties
            assert r[b] == int((negs[b] >= pos[b]).sum())
        for i, k in enumerate((1, 3, 5)):
            assert hits[b, i].item() == float(r[b] < k)
            assert abs(ndcg[b, i].item() - (1 / np.log2(r[b] + 2) if r[b] < k else 0.0)) < 1e-6


def test_debug_check_reports_updates_into_the_past():
    """memory_updater.py:25,41: feeding an earlier batch after a later one must raise the reference's AssertionError
    (opt-in here: the check synchronises the device)."""
    cfg = SyntheticConfig("t", 200, 20, 3000, 32, 1, 5, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    tgn = P.TGN(P.get_neighbor_finder(d, uniform=False), g.node_features, g.edge_features, DEV, n_layers=1, n_heads=2,
                dropout=0.0, use_memory=True, memory_dimension=32, message_function="identity", n_neighbors=5)
    tgn.debug_checks = True
    rs = np.random.RandomState(0)
    B = 30

    def run(s):
        neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)
        with torch.no_grad():
            tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B],
                                            d.edge_idxs[s:s + B], 5)
    run(2000)                       # messages at t ~ late
    run(2030)                       # persists them (last_update ~ late), stores new ones
    # a batch from the distant past touching the same graph: its messages carry old times; once such a message is
    # pending for a node whose last_update is later, the next call must refuse
    run(100)
    with pytest.raises(AssertionError, match="time in the past"):
        run(130)
        run(160)


def test_rank_metrics_against_reference_fixture():
    """pfo_rank_metrics vs the reference's recall_at_k / ndcg_at_k (fixture g6, generated by tools/make_golden.py from
    evaluation.py:11-21,122-128): embeddings are built so that the dot products ARE the fixture's scores."""
    g = load_golden("g6_eval_metrics")
    s = g["scores"]
    B, N = s.shape[0], s.shape[1] - 1
    D = 8
    emb = np.zeros((B * (2 + N), D), np.float32)
    emb[:B, 0] = 1.0                                              # source = e0: score = first component of the other side
    emb[B:2 * B, 0] = s[:, 0]
    emb[2 * B:, 0] = s[:, 1:].reshape(-1)
    rank, hits, ndcg = P.rank_metrics(torch.from_numpy(emb).to(DEV), B, N)
    r = rank.cpu().numpy()
    assert np.array_equal(r, g["n_greater"] + g["n_equal"])       # canonical tie policy: behind every negative that scores >= it
    free = g["n_equal"] == 0
    assert np.array_equal(r[free], g["pos_rank"][free])            # without ties: the reference's ranking position
    assert np.all(g["pos_rank"] <= r) and np.all(g["pos_rank"] >= g["n_greater"])
    assert np.array_equal(hits.cpu().numpy()[free].astype(np.float64), g["recall"][free])
    assert np.allclose(ndcg.cpu().numpy()[free], g["ndcg"][free], rtol=0, atol=1e-6)


def test_step_against_oracle_on_a_general_graph_with_self_loops_and_ties():
    """Not bipartite: arbitrary node pairs, self-loops (both row entries of a self-loop carry one edge id) and few distinct
    timestamps.  The run-merged layer-1 backward groups instances by (node, newest neighbour entry): this graph is where two
    instances of a node can share the newest edge id and still differ in their neighbour lists."""
    torch.manual_seed(77)
    rs = np.random.RandomState(8)
    n_nodes, E, D, K, L, H = 61, 1500, 16, 5, 2, 2
    src = rs.randint(1, n_nodes, E)
    dst = rs.randint(1, n_nodes, E)
    dst[::9] = src[::9]                                            # self-loops
    ts = np.sort(rs.randint(0, 60, E)).astype(np.float64)          # ~25 edges per timestamp
    eidx = np.arange(1, E + 1)
    node_feat, edge_feat = rs.rand(n_nodes, D), rs.randn(E + 1, 4)
    edge_feat[0] = 0
    nf = P.NeighborFinder.from_arrays(src, dst, eidx, ts, max_node_idx=n_nodes - 1)
    tgn = P.TGN(nf, node_feat, edge_feat, DEV, n_layers=L, n_heads=H, dropout=0.0, use_memory=True, memory_dimension=D,
                message_function="identity", n_neighbors=K)
    with torch.no_grad():
        tgn.time_encoder.w.bias.normal_(0, 0.3)
    onf = OracleNeighborFinder(*build_adjacency(src, dst, eidx, ts, max_node_idx=n_nodes - 1))
    names = [k for k in tgn.state_dict() if "layer_norm" not in k and not k.startswith("memory.")]
    ref = T.OracleTGN(onf, node_feat, edge_feat, {k: tgn.state_dict()[k].cpu().numpy() for k in names}, L, H, True)
    opt = P.FusedAdam(tgn, lr=1e-3)
    B = 48
    for step in range(3):
        s = 900 + step * B
        sb, db, tb, eb = src[s:s + B], dst[s:s + B], ts[s:s + B], eidx[s:s + B]
        neg = rs.randint(1, n_nodes, size=B * 3)
        ref.P = {k: tgn.state_dict()[k].detach().cpu().numpy().copy() for k in names}
        tgn.train(); opt.zero_grad()
        se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, K)
        rse, rde, rne = ref.compute_temporal_embeddings(sb, db, neg, tb, eb, K)
        emb = torch.cat([se, de, ne])
        assert relerr(emb.detach().cpu().numpy(), np.concatenate([rse, rde, rne])) < RTOL_EMB
        rgrads = _masked_bpr_backward(tgn, ref, emb, rse, rde, rne, B, K)       # see test_step_against_oracle
        tol, tol_time = RTOL_GRAD_ORACLE_L2, RTOL_GRAD_TIME
        for name, p in tgn.named_parameters():
            if name not in rgrads or np.abs(rgrads[name]).max() < 1e-7:
                continue
            r = rgrads[name].reshape(p.shape)
            if p.grad is None:                                      # step 0: no pending message, the GRU is not called
                assert name.startswith("memory_updater") and np.abs(r).max() == 0
                continue
            got = p.grad.cpu().numpy().astype(np.float64)
            e = np.linalg.norm(got - r) / (np.linalg.norm(r) + 1e-30)
            assert e < (tol_time if name.startswith("time_encoder") else tol), (step, name, e)
        assert relerr(tgn.memory.memory.cpu().numpy(), ref.memory) < RTOL_EMB
        assert np.array_equal(tgn.memory.last_update.cpu().numpy(), ref.last_update)
        opt.step()
