"""GPU tests added in round 3.

* SURVEY 8(f-4) end to end on the device: the reference's file layout (``time_feature[day][code] -> 30 prices`` pickle,
  portfolios as stock-code lists, ``map_item_id``) -> ``prices_from_time_feature`` -> ``day_indices`` ->
  ``pack_portfolios`` -> ``MVSampler.select`` lands on the g3 fixture (captured by executing main.py:192-304).
* RCCL: backend ``nccl`` initialised on the box, the flat gradient buffer all-reduced (world 1: proves the library loads and
  takes the buffer the data-parallel step hands it).
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]

import pfotgnrec_amd as P
from pfotgnrec_amd.mv_sampler import prices_from_time_feature, day_indices
from pfotgnrec_amd.rand_edge_sampler import pack_portfolios
from oracle import mv_select as omv

DEV = "cuda:0"


# ------------------------------------------------------------------ f-4: pickle layout -> MV selection == g3
@pytest.mark.parametrize("lam", [0.5, 0.1])
@pytest.mark.parametrize("keys", ["codes", "node_ids", "both"])
def test_file_layout_to_mv_selection_matches_reference(lam, keys):
    g = load_golden("g3_mv")
    pre = "lam%02d_" % int(lam * 10)
    upper_u = int(g["upper_u"])
    codes = [str(c) for c in g["codes"]]
    map_item_id = {c: j for j, c in enumerate(codes)}                      # map_item_id.pkl (main.py:89)
    # time_feature_future_{p}.pkl the way main.py:88,212-227 consumes it: day key -> {stock: 30 prices}; main.py:223 looks
    # portfolio members up by CODE, main.py:217/224 candidates by item NODE id - both kinds of keys may be present
    time_feature = {}
    for key, day in zip(g["day_keys"], g[pre + "day_idx"]):
        inner = {}
        for j, c in enumerate(codes):
            if keys in ("codes", "both"):
                inner[c] = g["prices"][day, j]
            if keys in ("node_ids", "both"):
                inner[upper_u + 1 + j] = g["prices"][day, j]
        time_feature[str(key)] = inner
    portfolios = [[c for c in row if c] or [""] for row in g["port_codes"].tolist()]   # ml_transaction.json "portfolio"
    # ---- ingest
    days, arr = prices_from_time_feature(time_feature, map_item_id, upper_u=upper_u)
    assert arr.shape == (len(days), len(codes), 30) and len(days) == len(set(str(k) for k in g["day_keys"]))
    port_idx, port_len = pack_portfolios(portfolios, map_item_id, width=g["port_idx"].shape[1])
    assert np.array_equal(port_len, g["port_len"])
    for b in range(len(port_len)):
        assert np.array_equal(port_idx[b, :port_len[b]], g["port_idx"][b, :port_len[b]])
    mvs = P.MVSampler(arr, upper_u, DEV, gamma=float(g["gamma"]), lambda_mv=lam, p_pos_num=1, p_neg_num=3,
                      day_of=lambda ts: day_indices(ts, days))              # str(ts)[:8], main.py:212
    # ---- selection on the device
    p_pos, p_neg, y, nr = mvs.select(g["dst"], g[pre + "neg"], g["ts"], port_idx, port_len, want_scores=True)
    assert np.allclose(y, g[pre + "y_mv"], rtol=1e-11, atol=0)             # fp64; summation order of the covariance
    assert np.array_equal(np.stack([omv.fuse_ranks(r, lam)[0] for r in y]), g[pre + "invest_rank"])
    assert np.array_equal(nr, g[pre + "new_rank"])
    cand = np.concatenate([g["dst"][:, None], g[pre + "neg"]], 1)
    n_tiefree = 0
    for b in range(len(cand)):
        order = omv.canonical_order(g[pre + "new_rank"][b])
        assert p_pos[b] == cand[b][order[0]] and np.array_equal(p_neg[3 * b:3 * b + 3], cand[b][order[-3:]])
        ref_order, nrb = g[pre + "order"][b], g[pre + "new_rank"][b]
        if not ((np.sum(nrb == nrb[ref_order[0]]) > 1) or any(np.sum(nrb == nrb[i]) > 1 for i in ref_order[-3:])):
            n_tiefree += 1                                                 # tie-free: the reference's own selection
            assert p_pos[b] == g[pre + "p_pos"][b] and np.array_equal(p_neg[3 * b:3 * b + 3], g[pre + "p_neg"][3 * b:3 * b + 3])
    assert n_tiefree > 0


# ------------------------------------------------------------------ RCCL takes the flat gradient buffer
_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import torch.distributed as dist
import pfotgnrec_amd as P
from pfotgnrec_amd.distributed import allreduce_flat_grad, init_from_env
from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
DEV = "cuda:0"
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1)
assert dist.get_backend() == "nccl"
cfg = SyntheticConfig("r3", 200, 20, 3000, 32, 2, 6, 2)
gr = make_graph(cfg, with_prices=False)
tgn = P.TGN(P.get_neighbor_finder(gr.data, False), gr.node_features, gr.edge_features, DEV, n_layers=2, n_heads=2,
            dropout=0.0, use_memory=True, memory_dimension=32, message_function="identity", n_neighbors=6)
d = gr.data
rs = np.random.RandomState(0)
for s in (1500, 1540):
    neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=40 * 3)
    tgn.zero_grad(set_to_none=True)
    se, de, ne = tgn.compute_temporal_embeddings(d.sources[s:s + 40], d.destinations[s:s + 40], neg, d.timestamps[s:s + 40],
                                                 d.edge_idxs[s:s + 40], 6)
    P.bpr_loss(torch.cat([se, de, ne]), 40, 3).backward()
before = tgn.flat_grad.clone()
assert float(before.abs().sum()) > 0
# world 1 through the REAL collective (the production helper returns early at world 1): RCCL is loaded, a communicator
# exists and the flat fp32 buffer goes through ncclAllReduce in place
dist.all_reduce(tgn.flat_grad, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
assert torch.equal(tgn.flat_grad, before)
# the two-bucket form on the side stream (distributed.allreduce_flat_grad_buckets) through the same communicator
from pfotgnrec_amd.distributed import allreduce_flat_grad_buckets
allreduce_flat_grad_buckets(tgn, 1, force=True)
torch.cuda.synchronize()
assert torch.equal(tgn.flat_grad, before)
t = torch.arange(8, device=DEV, dtype=torch.float32)
dist.broadcast(t, src=0)
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", flush=True)
"""


def test_rccl_world1_allreduce_of_flat_gradient():
    """A CHILD process (a hang or an RCCL load failure cannot take the suite down; it is killed after 5 minutes)."""
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    port = 29700 + (os.getpid() % 200)
    out = subprocess.run(["timeout", "-k", "10", "300", sys.executable, "-c", _RCCL_CHILD % REPO, str(port)], env=env,
                         capture_output=True)
    assert out.returncode == 0 and b"RCCL_OK" in out.stdout, out.stderr.decode()[-3000:]


# ------------------------------------------------------------------ deterministic backward
@pytest.mark.parametrize("uniform", [False, True])
def test_deterministic_backward_is_bitwise_reproducible(uniform):
    """``tgn.deterministic = True`` (pfo_tgn_batch.deterministic): the same step from the same state gives bit-identical
    gradients run after run - level-0 rows summed as 2^-40 fixed-point int64, time-encoder partials folded from per-workgroup
    slab rows - and agrees with the default (float-atomic) backward to fp32 rounding.  Most-recent sampling takes the
    shift-merged layer-1 kernel, uniform sampling the per-instance one."""
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    torch.manual_seed(3)
    cfg = SyntheticConfig("det", 400, 30, 8000, 64, 2, 10, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    tgn = P.TGN(P.get_neighbor_finder(d, uniform), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.1,
                use_memory=True, memory_dimension=64, message_function="identity", n_neighbors=10)
    with torch.no_grad():
        tgn.time_encoder.w.bias.normal_(0, 0.3)
    rs = np.random.RandomState(1)
    B = 96
    tgn.train()
    for s in (4000, 4096):                                              # populate memory and pending messages
        neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)
        with torch.no_grad():
            tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B], d.edge_idxs[s:s + B], 10)
    m = tgn.memory
    snap = [t.clone() for t in (m.memory.data, m.last_update.data, m.msg_table, m.msg_time, m.has_msg)]
    s = 4192
    batch = (d.sources[s:s + B], d.destinations[s:s + B], rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3),
             d.timestamps[s:s + B], d.edge_idxs[s:s + B])

    def grad(det):
        with torch.no_grad():
            for dst_t, src_t in zip((m.memory.data, m.last_update.data, m.msg_table, m.msg_time, m.has_msg), snap):
                dst_t.copy_(src_t)
        tgn.deterministic = det
        tgn._step = 1000                                                # same Philox stream position (dropout masks, draws)
        tgn.zero_grad(set_to_none=True)
        emb = torch.cat(tgn.compute_temporal_embeddings(*batch, 10))
        P.bpr_loss(emb, B, 3).backward()
        return tgn.flat_grad.clone()

    g1, g2, g3 = grad(True), grad(True), grad(True)
    assert torch.equal(g1, g2) and torch.equal(g2, g3)
    g0 = grad(False)
    D = 64
    den = g0[2 * D:].abs().max().item()
    assert (g1[2 * D:] - g0[2 * D:]).abs().max().item() <= 2e-5 * den       # same sums, different rounding order
    assert (g1[:2 * D] - g0[:2 * D]).abs().max().item() <= 3e-3 * g0[:2 * D].abs().max().item()   # time encoder: cancellation-heavy
    tgn.deterministic = False


# ------------------------------------------------------------------ next batch prepared beside this batch's backward
@pytest.mark.parametrize("use_memory", [True, False])
def test_prefetched_batches_give_the_same_steps(use_memory):
    """``TGN.prefetch`` (pfo_tgn_prepare: root assembly, frontier sampling, compaction, packed memory rows of the NEXT batch on
    a second stream, into a second workspace, issued right after the current forward) changes when that work runs, not what
    it computes: embeddings, gradients (deterministic mode: bitwise), parameters after Adam and the memory tables of a 5-step
    run are identical with and without it.  A prefetch whose batch does not come next is dropped."""
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    cfg = SyntheticConfig("pre", 300, 25, 6000, 64, 2, 8, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    B, q, K = 64, 3, 8
    rs = np.random.RandomState(5)
    starts = [3000 + 64 * i for i in range(5)]
    negs = [rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * q) for _ in starts]

    def run(prefetch):
        torch.manual_seed(11)
        tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.1,
                    use_memory=use_memory, memory_dimension=64, message_function="identity", n_neighbors=K)
        tgn.deterministic = True
        tgn.train()
        opt = torch.optim.Adam(tgn.parameters(), lr=1e-3)
        dev = lambda a, t: tgn._to_dev(a, t)
        batches = [(dev(d.sources[s:s + B], np.int32), dev(d.destinations[s:s + B], np.int32), dev(n, np.int32),
                    dev(d.timestamps[s:s + B], np.float64), dev(d.edge_idxs[s:s + B], np.int32)) for s, n in zip(starts, negs)]
        out = []
        for i, (s_, d_, n_, t_, e_) in enumerate(batches):
            emb, b = tgn.embed_device(s_, d_, [n_], [q], t_, e_, K)
            if prefetch and i + 1 < len(batches):
                s2, d2, n2, t2, e2 = batches[i + 1]
                with tgn.prefetching():
                    if i == 1:                                     # a batch that does NOT come next: dropped when the real one arrives
                        assert tgn.prefetch(s2, d2, [batches[0][2]], [q], t2, e2, K)
                    else:
                        assert tgn.prefetch(s2, d2, [n2], [q], t2, e2, K)
            P.bpr_step(tgn, emb, b, q)
            grad = tgn.flat_grad.clone()
            opt.step()
            opt.zero_grad(set_to_none=True)
            out.append((emb.detach().clone(), grad, tgn._flat.detach().clone()))
        mem = [t.clone() for t in (tgn.memory.memory.data, tgn.memory.msg_table, tgn.memory.msg_time)] if use_memory else []
        torch.cuda.synchronize()
        return out, mem

    a, mem_a = run(False)
    b_, mem_b = run(True)
    for (e0, g0, p0), (e1, g1, p1) in zip(a, b_):
        assert torch.equal(e0, e1) and torch.equal(g0, g1) and torch.equal(p0, p1)
    for x, y in zip(mem_a, mem_b):
        assert torch.equal(x, y)
