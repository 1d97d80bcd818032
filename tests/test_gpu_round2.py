"""GPU tests added in round 2 for the host-side contract of the TGN facade and the literal drop-in loop:

* several training forwards before one backward (main.py:171 BACKPROP_EVERY), an evaluation forward between a forward and
  its backward: every outstanding call owns its workspace;
* a neighbour finder built from a split whose largest id is below the node table's (main.py:95 leaves max_node_idx=None);
* dropout follows train()/eval(), autograd follows torch.is_grad_enabled();
* zero negatives;
* main.py:160-394 in shape: package imports swapped per INTEGRATION.md, the torch BPR expression, torch.optim.Adam,
  detach_memory(), set_neighbor_finder / direct attribute assignment.
"""
import numpy as np
import pytest
import torch

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]

import pfotgnrec_amd as P
from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph

DEV = "cuda:0"


def _setup(L=2, D=32, K=6, H=2, dropout=0.0, use_memory=True, seed=9, uniform=False):
    torch.manual_seed(seed)
    cfg = SyntheticConfig("r2", 300, 25, 5000, D, L, K, H)
    g = make_graph(cfg, with_prices=False)
    tgn = P.TGN(P.get_neighbor_finder(g.data, uniform), g.node_features, g.edge_features, DEV, n_layers=L, n_heads=H,
                dropout=dropout, use_memory=use_memory, memory_dimension=D, message_function="identity", n_neighbors=K)
    return cfg, g, tgn


def _batch(cfg, g, s, B, rs):
    d = g.data
    neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)
    return d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B], d.edge_idxs[s:s + B]


def _snapshot(tgn):
    m = tgn.memory
    return [t.clone() for t in (m.memory.data, m.last_update.data, m.msg_table, m.msg_time, m.has_msg)]


def _restore(tgn, snap):
    m = tgn.memory
    with torch.no_grad():
        for dst, src in zip((m.memory.data, m.last_update.data, m.msg_table, m.msg_time, m.has_msg), snap):
            dst.copy_(src)


def test_two_forwards_before_one_backward_and_eval_in_between():
    cfg, g, tgn = _setup()
    rs = np.random.RandomState(0)
    b1, b2, b3 = (_batch(cfg, g, s, 40, rs) for s in (2500, 2540, 2580))
    with torch.no_grad():
        tgn.compute_temporal_embeddings(*_batch(cfg, g, 2400, 40, rs), 6)          # populate memory / messages
    snap = _snapshot(tgn)

    def loss_of(batch):
        se, de, ne = tgn.compute_temporal_embeddings(*batch, 6)
        return P.bpr_loss(torch.cat([se, de, ne]), 40, 3)

    # reference: one backward per forward, gradients summed
    tgn.train()
    for p in tgn.parameters():
        p.grad = None
    l1 = loss_of(b1); l1.backward()
    l2 = loss_of(b2); l2.backward()
    want = tgn.flat_grad.clone()
    mem_after = _snapshot(tgn)
    # accumulated: two forwards, then ONE backward of the summed loss, with an evaluation forward in between
    _restore(tgn, snap)
    for p in tgn.parameters():
        p.grad = None
    l1 = loss_of(b1)
    l2 = loss_of(b2)
    tgn.eval()
    with torch.no_grad():
        bk = tgn.memory.backup_memory()
        tgn.compute_temporal_embeddings(*b3, 6)                                     # a validation pass (main.py:396-420)
        tgn.memory.restore_memory(bk)
    tgn.train()
    (l1 + l2).backward()
    got = tgn.flat_grad
    den = want.abs().max().item()
    assert (got - want).abs().max().item() / den < 2e-5            # float atomics: order-dependent in the last bits
    for a, b in zip(mem_after, _snapshot(tgn)):
        assert torch.equal(a, b)
    # the workspace of a finished call is not reusable for a second backward
    l = loss_of(b1)
    l.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already consumed"):
        l.backward()


def test_finder_smaller_than_node_table_is_padded_and_larger_is_refused():
    cfg, g, tgn = _setup(L=2, use_memory=True)
    d = g.data
    keep = np.maximum(d.sources, d.destinations) <= 310             # a "train split" that never sees items 311..325
    sub = P.Data(d.sources[keep], d.destinations[keep], d.timestamps[keep], d.edge_idxs[keep])
    small = P.get_neighbor_finder(sub, uniform=False)               # main.py:95: max_node_idx=None
    assert small.n_nodes < tgn.n_nodes
    full = P.get_neighbor_finder(sub, uniform=False, max_node_idx=tgn.n_nodes - 1)
    rs = np.random.RandomState(1)
    s, B = 3000, 32
    sb, db, tb, eb = d.sources[s:s + B], d.destinations[s:s + B], d.timestamps[s:s + B], d.edge_idxs[s:s + B]
    neg = rs.randint(311, 326, size=B * 3)                          # roots the small finder has no row for
    outs = []
    tgn.eval()
    for nf in (small, full):
        tgn.memory.__init_memory__()
        tgn.set_neighbor_finder(nf)
        with torch.no_grad():
            outs.append(torch.cat(tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, 6)))
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
    # ids beyond the node table: the reference raises IndexError from its feature lookup
    with pytest.raises(IndexError):
        tgn.compute_temporal_embeddings(sb, db, np.full(B * 3, tgn.n_nodes), tb, eb, 6)
    big = P.NeighborFinder.from_arrays(np.array([1, tgn.n_nodes + 3]), np.array([305, 306]), np.array([1, 2]), np.array([1.0, 2.0]))
    tgn.embedding_module.neighbor_finder = big                      # main.py:427 assigns the attribute directly
    with pytest.raises(ValueError, match="node ids"):
        tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, 6)


def test_tgn_sees_edges_appended_to_its_finder():
    """NeighborFinder.append replaces the device arrays: a TGN holding the finder must pick the new adjacency up."""
    cfg, g, tgn = _setup(L=1, use_memory=False)
    d = g.data
    half = 2500
    nf = P.NeighborFinder.from_arrays(d.sources[:half], d.destinations[:half], d.edge_idxs[:half], d.timestamps[:half],
                                      max_node_idx=tgn.n_nodes - 1)
    full = P.NeighborFinder.from_arrays(d.sources[:4000], d.destinations[:4000], d.edge_idxs[:4000], d.timestamps[:4000],
                                        max_node_idx=tgn.n_nodes - 1)
    rs = np.random.RandomState(5)
    batch = _batch(cfg, g, 4000, 24, rs)
    tgn.eval()
    tgn.set_neighbor_finder(nf)
    with torch.no_grad():
        before = torch.cat(tgn.compute_temporal_embeddings(*batch, 6))
        nf.append(d.sources[half:4000], d.destinations[half:4000], d.edge_idxs[half:4000], d.timestamps[half:4000], device=DEV)
        after = torch.cat(tgn.compute_temporal_embeddings(*batch, 6))
        tgn.set_neighbor_finder(full)
        want = torch.cat(tgn.compute_temporal_embeddings(*batch, 6))
    assert torch.equal(after, want) and not torch.equal(before, after)


def test_dropout_follows_train_flag_and_autograd_follows_grad_mode():
    cfg, g, tgn = _setup(L=1, dropout=0.5, use_memory=False)
    rs = np.random.RandomState(2)
    batch = _batch(cfg, g, 2600, 24, rs)

    def run():
        tgn._step = 3
        return torch.cat(tgn.compute_temporal_embeddings(*batch, 6))
    tgn.eval()
    e_eval = run()                                                   # grad mode on, eval(): no dropout, differentiable
    assert e_eval.requires_grad and e_eval.grad_fn is not None
    e_eval.sum().backward()
    assert tgn.flat_grad is not None and tgn.flat_grad.abs().sum().item() > 0
    with torch.no_grad():
        e_eval_ng = run()
    assert torch.equal(e_eval.detach(), e_eval_ng)
    tgn.train()
    with torch.no_grad():
        e_train_ng = run()                                           # train() under no_grad: dropout IS applied (the reference's nn.Dropout)
    assert not e_train_ng.requires_grad
    assert not torch.equal(e_train_ng, e_eval_ng)
    e_train = run()
    assert torch.equal(e_train.detach(), e_train_ng)                 # same Philox stream with and without autograd


def test_zero_negatives_and_zero_neighbors():
    cfg, g, tgn = _setup(L=1)
    rs = np.random.RandomState(3)
    sb, db, neg, tb, eb = _batch(cfg, g, 2700, 16, rs)
    tgn.eval()
    with torch.no_grad():
        se, de, ne = tgn.compute_temporal_embeddings(sb, db, np.zeros(0, np.int64), tb, eb, 6)
        assert se.shape == (16, 32) and de.shape == (16, 32) and ne.shape == (0, 32)
        tgn.memory.__init_memory__()
        se2, de2, ne2 = tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, 6)
        assert ne2.shape == (48, 32)
        tgn.memory.__init_memory__()
        se0, _, _ = tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, 0)     # utils.py:175: one all-padding column
        assert torch.isfinite(se0).all()


def _main_loop_step(tgn, optimizer, d, cfg, k, negatives_batch, literal):
    """One iteration of main.py:160-394 (baseline branch, :345-394) with the package's TGN in the reference's place."""
    BATCH_SIZE, NUM_NEG_TRAIN, NUM_NEIGHBORS = 40, 3, 6
    optimizer.zero_grad()                                            # main.py:169
    start_idx = 2500 + k * BATCH_SIZE
    end_idx = start_idx + BATCH_SIZE
    sources_batch, destinations_batch = d.sources[start_idx:end_idx], d.destinations[start_idx:end_idx]
    edge_idxs_batch, timestamps_batch = d.edge_idxs[start_idx:end_idx], d.timestamps[start_idx:end_idx]
    tgn = tgn.train()                                                # main.py:309
    source_embedding, destination_embedding, negative_embedding = tgn.compute_temporal_embeddings(
        sources_batch, destinations_batch, negatives_batch.flatten(), timestamps_batch, edge_idxs_batch, NUM_NEIGHBORS)
    if literal:
        bs = source_embedding.shape[0]                               # main.py:364-381, verbatim
        source_embedding = source_embedding.view(bs, 1, -1)
        destination_embedding = destination_embedding.view(bs, 1, -1)
        negative_embedding = negative_embedding.view(bs, NUM_NEG_TRAIN, -1)
        pos_scores = torch.sum(source_embedding * destination_embedding, dim=2)
        neg_scores = torch.matmul(source_embedding, negative_embedding.transpose(1, 2)).squeeze()
        score_diff = pos_scores - neg_scores
        score_diff_mean = torch.mean(score_diff, dim=1)
        log_and_sigmoid = torch.log(torch.sigmoid(score_diff_mean))
        loss = -torch.mean(log_and_sigmoid)
    else:
        loss = P.bpr_loss(torch.cat([source_embedding, destination_embedding, negative_embedding]), BATCH_SIZE, NUM_NEG_TRAIN)
    loss.backward()                                                  # main.py:388
    grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in tgn.named_parameters() if p.requires_grad}
    optimizer.step()                                                 # main.py:389
    tgn.memory.detach_memory()                                       # main.py:394
    return float(loss.detach()), grads


def test_literal_main_loop_with_torch_bpr_and_torch_adam():
    """main.py:160-394 with the package swapped in: the reference's own torch expressions for the BPR
    loss (main.py:364-381), torch.optim.Adam(tgn.parameters()) (main.py:123), loss.backward(), optimizer.step(),
    tgn.memory.detach_memory() (main.py:394), set_neighbor_finder / direct attribute assignment (main.py:156, 427) -
    against the fused path (pfo_bpr_loss + FusedAdam).  The two loops run in lockstep: before every step the literal
    model takes over the fused model's parameters, memory and Adam moments, so each of the four steps is compared from
    an identical state (free-running trajectories cannot be: Adam turns rounding-level gradient differences into +-lr
    moves of the time-encoder frequencies, SURVEY 7 hard part 5).  Step 0 starts from an empty memory: the GRU tensors
    get NO gradient (None, not zero) and both optimizers must skip them."""
    models = []
    for literal in (True, False):
        cfg, g, tgn = _setup(L=2, D=32, K=6, seed=21)
        d = g.data
        train_ngh_finder = P.get_neighbor_finder(d, uniform=False)
        optimizer = torch.optim.Adam(tgn.parameters(), lr=1e-3) if literal else P.FusedAdam(tgn, lr=1e-3)
        tgn.memory.__init_memory__()                                 # main.py:153
        tgn.set_neighbor_finder(train_ngh_finder)                    # main.py:156
        models.append((tgn, optimizer))
    (A, optA), (Bm, optB) = models
    namesA, namesB = dict(A.named_parameters()), dict(Bm.named_parameters())
    viewsB = {p_: (off, n) for p_, off, n, _ in Bm._views}
    rs = np.random.RandomState(4)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    for k in range(4):
        if k > 0:                                                    # lockstep: A <- B
            with torch.no_grad():
                A.flat_parameters.copy_(Bm.flat_parameters)
                A.memory.restore_memory(Bm.memory.backup_memory())
            for n_, pB in namesB.items():
                if pB not in viewsB or pB not in optB._steps:
                    continue
                off, cnt = viewsB[pB]
                optA.state[namesA[n_]] = {"step": torch.tensor(float(optB._steps[pB])),
                                          "exp_avg": optB._m[off:off + cnt].view(pB.shape).clone(),
                                          "exp_avg_sq": optB._v[off:off + cnt].view(pB.shape).clone()}
        before = Bm.flat_parameters.clone()
        m_before = None if optB._m is None else optB._m.clone()
        v_before = None if optB._v is None else optB._v.clone()
        steps_before = optB._steps.copy()
        negatives_batch = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=(40, 3))
        la, ga = _main_loop_step(A, optA, d, cfg, k, negatives_batch, True)
        lb, gb = _main_loop_step(Bm, optB, d, cfg, k, negatives_batch, False)
        assert abs(la - lb) < 2e-6 * max(1.0, abs(lb)), (k, la, lb)
        for n_ in gb:
            if "layer_norm" in n_:                                   # constructed, never applied (memory_updater.py:14)
                assert ga[n_] is None and gb[n_] is None
                continue
            assert (ga[n_] is None) == (gb[n_] is None), (k, n_)
            gru = n_.startswith("memory_updater.memory_updater.")
            assert (gb[n_] is None) == (gru and k == 0), (k, n_)     # memory_updater.py:38-40: no message, no GRU call
            if gb[n_] is None:
                assert torch.equal(namesA[n_].detach(), namesB[n_].detach())          # skipped by both optimizers
                continue
            den = gb[n_].abs().max().item()
            if den < 1e-12:
                continue
            tol = 2e-3 if n_.startswith("time_encoder") else 2e-5
            assert (ga[n_] - gb[n_]).abs().max().item() / den < tol, (k, n_)
            # post-step parameters: identical Adam from identical state, up to the amplified gradient difference
            off, cnt = viewsB[namesB[n_]]
            t = steps_before.get(namesB[n_], 0) + 1
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            v0 = torch.zeros(cnt, device=DEV) if v_before is None else v_before[off:off + cnt]
            gB = gb[n_].reshape(-1)
            v_new = b2 * v0 + (1 - b2) * gB * gB
            bound = lr * (1 - b1) / bc1 * (ga[n_].reshape(-1) - gB).abs() / ((v_new / bc2).sqrt() + eps)
            diff = (namesA[n_].detach().reshape(-1) - namesB[n_].detach().reshape(-1)).abs()
            slack = 4e-7 * namesB[n_].detach().reshape(-1).abs().clamp(min=1.0) + 2e-3 * lr
            assert bool((diff <= 1.5 * bound + slack).all()), (k, n_, diff.max().item())
        assert torch.equal(A.memory.last_update, Bm.memory.last_update)
        assert (A.memory.memory - Bm.memory.memory).abs().max().item() < 1e-5
        assert not torch.equal(before, Bm.flat_parameters)
    assert optB._steps[namesB["memory_updater.memory_updater.weight_ih"]] == 3 and optB._steps[namesB["time_encoder.w.weight"]] == 4
    A.embedding_module.neighbor_finder = train_ngh_finder            # main.py:427
    assert A.neighbor_finder is train_ngh_finder


def test_torch_library_ops_match_the_ctypes_wrappers():
    """north_star: "PyTorch-ROCm custom ops".  The stateless entry points are registered with torch.library
    (pfotgnrec_amd/ops.py): same results as the direct wrappers, autograd through the BPR op, no CPU kernel."""
    g = load_golden_r2("g1_sampler")
    nf = P.NeighborFinder.from_arrays(g["a_src"], g["a_dst"], g["a_eidx"], g["a_ts"])
    indptr, nbr, eidx, ts = nf.device_arrays(torch.device(DEV))
    q = torch.from_numpy(g["a_q_nodes"].astype(np.int32)).to(DEV)
    qt = torch.from_numpy(g["a_q_ts"].astype(np.float64)).to(DEV)
    o_nbr, o_eidx, o_et = torch.ops.pfotgn.tnbr_sample(indptr, nbr, eidx, ts, q, qt, 10)
    assert np.array_equal(o_nbr.cpu().numpy(), g["a_K10_nbr"]) and np.array_equal(o_et.cpu().numpy(), g["a_K10_et"])
    t = torch.rand(7, 5, device=DEV) * 1e4
    w, b = torch.rand(16, 1, device=DEV), torch.rand(16, device=DEV)
    assert torch.equal(torch.ops.pfotgn.time_encode(t, w, b), P.time_encode(t, w.reshape(-1), b))
    emb = torch.randn(5 * 12, 16, device=DEV, requires_grad=True)
    l1 = P.ops.bpr_loss(emb, 12, 3)
    l1.backward()
    g1 = emb.grad.clone(); emb.grad = None
    l2 = P.bpr_loss(emb, 12, 3)
    l2.backward()
    assert torch.equal(l1.detach(), l2.detach()) and torch.equal(g1, emb.grad)
    with pytest.raises(Exception):
        torch.ops.pfotgn.time_encode(torch.zeros(3), torch.ones(4, 1), torch.zeros(4))       # no CPU implementation


def load_golden_r2(name):
    from conftest import load_golden
    return load_golden(name)


@pytest.mark.parametrize("ours", [False, True])
def test_graphed_step_equals_eager_step(ours):
    """pfotgnrec_amd/graph.py: the whole training step (candidate draw, [MV selection,] sampling, memory, attention, BPR,
    backward, Adam, state update) replayed from ONE captured HIP graph equals the same step queued kernel by kernel - same
    device-side stream positions, so the same dropout masks and negatives.  Lockstep (the replaying model takes over the
    eager model's state before every step): free-running trajectories separate through Adam (SURVEY 7 hard part 5)."""
    from pfotgnrec_amd.rand_edge_sampler import item_availability, DeviceNegativeSampler
    lr = 1e-3
    sides = []
    for _ in range(2):
        torch.manual_seed(31)
        cfg = SyntheticConfig("gs", 300, 25, 5000, 32, 2, 6, 2)
        g = make_graph(cfg, with_prices=ours)
        d = g.data
        tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.1,
                    use_memory=True, memory_dimension=32, message_function="identity", n_neighbors=6)
        opt = P.FusedAdam(tgn, lr=lr)
        sampler = DeviceNegativeSampler(item_availability(d.destinations, g.upper_u, cfg.n_items), g.upper_u, DEV, seed=1)
        mvs = P.MVSampler(g.prices, g.upper_u, DEV, gamma=2.0, lambda_mv=0.5, p_pos_num=1, p_neg_num=3) if ours else None
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(DEV)
        B = 32
        arrs = (t(d.sources, np.int32), t(d.destinations, np.int32), t(d.timestamps, np.float64), t(d.edge_idxs, np.int32),
                t(g.portfolio_idx, np.int32), t(g.portfolio_len, np.int32), t(g.day_of(d.timestamps), np.int32))
        gs = P.GraphedTrainStep(tgn, opt, sampler, B, 6, n_neg=3, port_width=arrs[4].shape[1], mv_sampler=mvs)
        sides.append((tgn, opt, gs, arrs))
    batch = lambda arrs, s: tuple(a[s:s + B] for a in arrs[:6]) + ((arrs[6][s:s + B],) if ours else (None,))
    (tA, oA, gA, aA), (tB, oB, gB, aB) = sides
    gA.capture(*batch(aA, 2400), warmup=2)                     # two eager steps + the capture (which executes nothing)
    gB.capture(*batch(aB, 2400), warmup=2)
    assert gA.graph is not None
    for k in range(4):
        with torch.no_grad():                                  # lockstep: A <- B (parameters, moments, memory, counters)
            tA.flat_parameters.copy_(tB.flat_parameters)
            oA._m.copy_(oB._m); oA._v.copy_(oB._v)
            tA.memory.restore_memory(tB.memory.backup_memory()); tA.memory._any_msg = True
            gA.rng_pos.copy_(gB.rng_pos); gA.adam_t.copy_(gB.adam_t)
        before = tB.flat_parameters.clone()
        la = float(gA(*batch(aA, 2432 + k * B)))               # ONE graph launch
        lb = float(gB.eager(*batch(aB, 2432 + k * B)))         # ~85 launches
        assert abs(la - lb) <= 2e-6 * max(1.0, abs(lb)), (k, la, lb)
        ga, gb = tA.flat_grad, tB.flat_grad
        assert (ga[64:] - gb[64:]).abs().max().item() <= 2e-5 * gb[64:].abs().max().item()
        assert (ga[:64] - gb[:64]).abs().max().item() <= 3e-3 * gb[:64].abs().max().item()      # time encoder (D = 32: w, b)
        dp = (tA.flat_parameters - tB.flat_parameters).abs()
        assert dp.max().item() <= 2.1 * lr                     # Adam: a last-bit gradient difference moves a weight by at most ~lr
        assert (dp > 1e-5).float().mean().item() < 2e-3        # ... and only where the gradient is at noise level
        assert not torch.equal(before, tB.flat_parameters)
        assert torch.equal(tA.memory.last_update, tB.memory.last_update) and torch.equal(tA.memory.msg_time, tB.memory.msg_time)
        assert (tA.memory.memory - tB.memory.memory).abs().max().item() < 1e-5
    assert gA.finish() == 4 and gB.finish() == 4
    assert oA._steps[dict(tA.named_parameters())["time_encoder.w.weight"]] == 6
