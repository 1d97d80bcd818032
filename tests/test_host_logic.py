"""CPU tests of the host-side mirrors: CSR build, portfolio packing, state_dict naming, unsupported options."""
import numpy as np
import pytest

import pfotgnrec_amd as P
from pfotgnrec_amd.neighbor_finder import build_csr
from pfotgnrec_amd.rand_edge_sampler import item_availability, pack_portfolios
from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph, CONFIGS
from conftest import load_golden
from oracle.neighbor_finder import build_adjacency


def test_csr_build_matches_oracle_on_golden_graphs():
    g = load_golden("g1_sampler")
    for p in ("a", "b"):
        got = build_csr(g[p + "_src"], g[p + "_dst"], g[p + "_eidx"], g[p + "_ts"])
        ref = build_adjacency(g[p + "_src"], g[p + "_dst"], g[p + "_eidx"], g[p + "_ts"])
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
        assert got[1].dtype == np.int32 and got[3].dtype == np.float64


def test_adj_list_constructor_equals_array_constructor():
    g = make_graph(SyntheticConfig("t", 30, 8, 300, 8, 1, 4, 2), with_prices=False)
    d = g.data
    adj = [[] for _ in range(g.n_nodes)]
    for s, t, e, ts in zip(d.sources, d.destinations, d.edge_idxs, d.timestamps):       # utils/utils.py:119-125
        adj[s].append((t, e, ts)); adj[t].append((s, e, ts))
    a = P.NeighborFinder(adj)
    b = P.get_neighbor_finder(d, False, max_node_idx=g.n_nodes - 1)
    for x, y in zip((a.indptr, a.nbr, a.eidx, a.ts), (b.indptr, b.nbr, b.eidx, b.ts)):
        assert np.array_equal(x, y)


def test_portfolio_packing_drops_empty_code():
    m = {"000001": 0, "000002": 1, "000003": 2}
    idx, ln = pack_portfolios([[""], ["000003", "000001"], ["000002"]], m)
    assert ln.tolist() == [0, 2, 1] and idx[1, :2].tolist() == [2, 0] and idx[0, 0] == -1
    av = item_availability(np.array([12, 10, 10]), 9, 3)
    assert av.tolist() == [1, 0, 1]


def test_state_dict_uses_reference_names():
    g = make_graph(SyntheticConfig("t", 30, 8, 300, 16, 2, 4, 2), with_prices=False)
    tgn = P.TGN(P.get_neighbor_finder(g.data, False), g.node_features, g.edge_features, "cpu", n_layers=2, n_heads=2,
                use_memory=True, memory_dimension=16, message_function="identity")
    ref = load_golden("g5_step_L2_mem")
    ref_names = {k[len("s2_sd_"):] for k in ref.files if k.startswith("s2_sd_")}
    assert ref_names <= set(tgn.state_dict().keys())
    for k in ref_names:
        if not k.startswith("memory."):              # memory tables are sized by the graph
            assert tuple(tgn.state_dict()[k].shape) == ref["s2_sd_" + k].shape, k
    # parameters are views of ONE flat buffer (single all-reduce, single Adam kernel)
    flat = tgn.flat_parameters
    for p in tgn.hot_parameters():
        assert p.data_ptr() >= flat.data_ptr() and p.data_ptr() < flat.data_ptr() + flat.numel() * 4
    # time-encoder initialisation (time_encoding.py:13-15)
    assert np.allclose(tgn.time_encoder.w.weight.detach().numpy().ravel(), 1 / 10 ** np.linspace(0, 9, 16), rtol=1e-6)


def test_unsupported_options_raise_like_the_reference_factories():
    g = make_graph(SyntheticConfig("t", 30, 8, 300, 8, 1, 4, 2), with_prices=False)
    nf = P.get_neighbor_finder(g.data, False)
    with pytest.raises(ValueError):
        P.TGN(nf, g.node_features, g.edge_features, "cpu", embedding_module_type="graph_sum")
    with pytest.raises(ValueError):
        P.TGN(nf, g.node_features, g.edge_features, "cpu", use_memory=True, memory_dimension=8, message_function="identity",
              aggregator_type="mean")
    with pytest.raises(ValueError):
        P.TGN(nf, g.node_features, g.edge_features, "cpu", use_memory=True, memory_dimension=8, message_function="identity",
              memory_updater_type="rnn")


def test_neighbor_finder_attribute_paths():
    """main.py:156/405 use set_neighbor_finder, main.py:427 assigns embedding_module.neighbor_finder directly."""
    g = make_graph(SyntheticConfig("t", 30, 8, 300, 8, 1, 4, 2), with_prices=False)
    a, b = P.get_neighbor_finder(g.data, False), P.get_neighbor_finder(g.data, True)
    tgn = P.TGN(a, g.node_features, g.edge_features, "cpu", n_layers=1, use_memory=False)
    assert tgn.neighbor_finder is a
    tgn.set_neighbor_finder(b)
    assert tgn.neighbor_finder is b and tgn.embedding_module.neighbor_finder is b
    tgn.embedding_module.neighbor_finder = a
    assert tgn.neighbor_finder is a


def test_time_statistics_match_reference_loop():
    g = make_graph(SyntheticConfig("t", 30, 8, 500, 8, 1, 4, 2), with_prices=False)
    d = g.data
    last_s, last_d, gs, gd = {}, {}, [], []
    for s, t, ts in zip(d.sources, d.destinations, d.timestamps):                      # utils/data.py:75-99
        gs.append(ts - last_s.get(s, 0)); gd.append(ts - last_d.get(t, 0)); last_s[s] = ts; last_d[t] = ts
    got = P.compute_time_statistics(d.sources, d.destinations, d.timestamps)
    assert np.allclose(got, (np.mean(gs), np.std(gs), np.mean(gd), np.std(gd)))


def test_synthetic_configs_match_baseline_json():
    c2 = CONFIGS["C2"]
    assert (c2.n_users, c2.n_items, c2.n_edges, c2.dim, c2.n_layers, c2.n_neighbors) == (50000, 500, 1000000, 172, 2, 20)
    assert CONFIGS["C5"].use_memory is False and CONFIGS["C5"].uniform and CONFIGS["C5"].n_heads == 4


def test_get_data_reads_the_reference_file_format(tmp_path):
    """utils/data.py:18-72 on files written the way utils/preprocess_data.py writes them."""
    import pandas as pd
    g = make_graph(SyntheticConfig("t", 40, 8, 600, 8, 1, 4, 2), with_prices=False)
    d = g.data
    base = tmp_path / "period_30"
    base.mkdir()
    df = pd.DataFrame({"u": d.sources, "i": d.destinations, "ts": d.timestamps, "label": np.zeros(len(d.sources), int),
                       "idx": d.edge_idxs, "portfolio": list(d.portfolios)})
    df.to_json(base / "ml_transaction.json")
    np.save(base / "ml_transaction.npy", g.edge_features)
    np.save(base / "ml_transaction_node.npy", g.node_features)
    nf, ef, full, train, val, test, upper_u = P.get_data("transaction", "30", root=str(tmp_path))
    assert upper_u == d.sources.max() and full.n_interactions == 600
    assert train.n_interactions + val.n_interactions + test.n_interactions == 600
    v, t = np.quantile(d.timestamps, [0.8, 0.9])
    assert train.timestamps.max() <= v < val.timestamps.min() and val.timestamps.max() <= t < test.timestamps.min()
    assert np.array_equal(full.sources, d.sources) and list(full.portfolios[5]) == list(d.portfolios[5])
    assert np.array_equal(ef, g.edge_features)


def test_price_ingest_matches_reference_expressions():
    """the reference's price pickle (day -> stock -> 30 prices; main.py:88, 212-227) packed into the dense
    [day, item, 30] tensor the MV kernel consumes; features = the reference's own np.log(p[1:] / p[:-1]) (fixture g7)."""
    from conftest import load_golden
    from pfotgnrec_amd.mv_sampler import prices_from_time_feature, log_returns, day_indices
    g = load_golden("g7_price_ingest")
    days, codes, prices = [str(x) for x in g["days"]], [str(x) for x in g["codes"]], g["prices"]
    upper_u = 100
    # the dict the way the pickle holds it: stock codes for portfolio lookups AND item node ids for candidate lookups
    tf = {dk: {} for dk in days}
    for i, dk in enumerate(days):
        for j, c in enumerate(codes):
            tf[dk][c] = prices[i, j]
            tf[dk][upper_u + 1 + j] = prices[i, j]
    map_item_id = {c: j for j, c in enumerate(codes)}
    got_days, arr = prices_from_time_feature(tf, map_item_id, upper_u=upper_u)
    assert got_days == sorted(days) and np.array_equal(arr, prices)
    only_ids = {dk: {k: v for k, v in tf[dk].items() if not isinstance(k, str)} for dk in days}
    assert np.array_equal(prices_from_time_feature(only_ids, map_item_id, upper_u=upper_u)[1], prices)
    ret = log_returns(arr)                                           # what MVSampler uploads
    day = day_indices(g["ts_batch"], got_days)                       # str(ts)[:8]
    feats, mus, k = g["features"], g["mus"], 0
    for b in range(len(day)):
        for q in range(int(g["cand_len"][b])):
            item = int(g["cand_idx"][b, q])
            assert np.array_equal(ret[day[b], item], feats[k])        # bit-exact fp64
            assert np.mean(ret[day[b], item]) == mus[k]
            k += 1
    assert k == len(feats)
    with pytest.raises(KeyError):
        day_indices([202402010000], got_days)


def test_eval_ranking_policy_against_reference_metrics():
    """evaluation.py:114-145 (fixture g6: rankings and recall / NDCG from the reference's own functions).  Canonical tie
    policy here: the positive ranks behind every negative that scores >= it.  Without ties that IS the
    reference's ranking; with ties the reference's position (np.argsort is not stable) lies in [n_greater, n_greater +
    n_equal] and ours is that interval's upper end."""
    from conftest import load_golden
    g = load_golden("g6_eval_metrics")
    s = g["scores"]
    rank = (s[:, 1:] >= s[:, :1]).sum(1)
    assert np.array_equal(rank, g["n_greater"] + g["n_equal"])
    free = g["n_equal"] == 0
    assert free.sum() >= 40 and np.array_equal(rank[free], g["pos_rank"][free])
    assert np.all((g["pos_rank"] >= g["n_greater"]) & (g["pos_rank"] <= rank))
    for i, k in enumerate(g["topk"]):
        rec = (rank < k).astype(np.float64)
        nd = np.where(rank < k, 1.0 / np.log2(rank + 2.0), 0.0)
        assert np.array_equal(rec[free], g["recall"][free, i])
        assert np.allclose(nd[free], g["ndcg"][free, i], rtol=0, atol=1e-12)


def test_rand_edge_sampler_availability_cache_follows_content():
    """the per-batch `np.unique(dst_list)` of utils.py:73 is cached on the array OBJECT (held, so its id cannot be
    recycled) plus a content fingerprint - never on `id()` alone."""
    import pfotgnrec_amd as P
    from pfotgnrec_amd.rand_edge_sampler import item_availability
    map_item_id = {"%06d" % (i + 1): i for i in range(10)}
    upper_u = 100
    src = np.array([1, 2, 3])
    ports = [[""], ["000001"], ["000002", "000003"]]
    a = np.array([101, 102, 103, 101], np.int64)
    s1 = P.RandEdgeSampler(src, a, ports, upper_u, map_item_id)
    assert s1.item_avail.tolist() == item_availability(a, upper_u, 10).tolist() == [1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
    s1b = P.RandEdgeSampler(src, a, ports, upper_u, map_item_id)
    assert s1b.item_avail is s1.item_avail                               # same long-lived array: one bitmap
    for _ in range(20):                                                   # temporaries of equal length (ids get recycled)
        b = np.array([105, 106, 107, 108], np.int64) + (_ % 2)
        s2 = P.RandEdgeSampler(src, b, ports, upper_u, map_item_id)
        assert s2.item_avail.tolist() == item_availability(b, upper_u, 10).tolist()
        del b
    a[0] = 110                                                            # in-place edit of the cached array
    s3 = P.RandEdgeSampler(src, a, ports, upper_u, map_item_id)
    assert s3.item_avail.tolist() == item_availability(a, upper_u, 10).tolist()
    assert s3.port_len.tolist() == [0, 1, 2]                              # '' dropped (utils.py:76)


# ------------------------------------------------------------------ the N-GPU record of bench.py
def _bench():
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module("bench")


def test_synthetic_eight_gpu_line_carries_the_fields_the_driver_reads():
    """No 8-GPU node is available to this build: the line a rank-path run prints is assembled here from synthetic numbers
    through the SAME helpers bench.main() uses (multi_gpu_line_fields, strong_scaling_entry) and checked against the schema
    (check_line_schema): collective / compute split, the all-reduce form, the C4 strong-scaling case with its one-GPU
    reference and efficiency."""
    b = _bench()
    for mode in ("single", "buckets", "fused", "fused_buckets"):
        line = {"metric": "interactions/sec (TGN fwd+BPR step)", "value": 8 * 360000.0, "unit": "interactions/s", "n_gpus": 8,
                "steps": 20, "warmup": 5, "ms_per_step": 1.42, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "C2 ...", "global_batch": 4096, "parallelism": "dp8", "allreduce": mode,
                           "collective": "rccl all-reduce of the flat fp32 gradient, world 8"}}
        line["config"].update(b.multi_gpu_line_fields(1.42, 0.11, 5, mode))
        line["secondary"] = {"strong_scaling_C4": b.strong_scaling_entry(1.9e6, 2.15, 8, "C4 ...", 4096, 0.12, 4.1e5, 9.99)}
        assert b.check_line_schema(line, 8) == []
        cfg = line["config"]
        if mode.startswith("fused"):          # the collective sits on the side stream: the caller's stream's step is all compute
            assert not cfg["collective_on_callers_stream"] and cfg["compute_ms_per_step"] == 1.42
        else:
            assert cfg["collective_on_callers_stream"] and abs(cfg["compute_ms_per_step"] + 0.11 - 1.42) < 1e-9
        eff = line["secondary"]["strong_scaling_C4"]["efficiency_vs_one_gpu"]
        assert abs(eff - 1.9e6 / (8 * 4.1e5)) < 1e-3
    # a line that lost a field is caught
    del line["config"]["collective_ms_per_step"]
    del line["secondary"]["strong_scaling_C4"]["one_gpu_reference"]
    miss = b.check_line_schema(line, 8)
    assert "config.collective_ms_per_step" in miss and "secondary.strong_scaling_C4.one_gpu_reference" in miss
    # one GPU: no collective fields asked for
    assert b.check_line_schema({k: line[k] for k in line if k != "secondary"} | {"n_gpus": 1, "config": {}}, 1) == []


def test_predicted_ring_time_of_the_gradient_all_reduce():
    """The stub collective of --emulate-ranks spins for this long on the side stream: 6.2 MB of fp32 gradients, ring over xGMI
    (2 (N-1)/N of the buffer per link direction at ~153 GB/s + 2 (N-1) hops)."""
    b = _bench()
    n_bytes = 1_551_000 * 4
    assert b.predicted_ring_allreduce_us(n_bytes, 1) == 0.0
    t2, t4, t8 = (b.predicted_ring_allreduce_us(n_bytes, w) for w in (2, 4, 8))
    assert 40 < t2 < t4 < t8 < 200               # DESIGN.md 6 quotes 70-110 us for the exchange at 2-8 ranks
    assert abs(t8 - (2 * 7 / 8 * n_bytes / 153e3 + 14 * 6.0)) < 1e-6


def test_portfolios_packed_once_for_slices_of_one_long_lived_array():
    """main.py:186 slices ``train_data.portfolios`` every batch: the whole array is packed once and a batch is a slice of the packed
    form (rand_edge_sampler.packed_portfolios_of) - identical to packing the slice itself, also for the [''] rows, with a wider
    pad; an in-place edit of the dataset that changes a list's length is caught; small arrays and copies are packed directly."""
    from pfotgnrec_amd.rand_edge_sampler import pack_portfolios, packed_portfolios_of, _PORT_CACHE
    m = {("%06d" % (i + 1)): i for i in range(60)}
    rs = np.random.RandomState(3)
    N = 5000
    arr = np.empty(N, dtype=object)
    for r in range(N):
        L = rs.randint(0, 8)
        arr[r] = [("%06d" % (j + 1)) for j in rs.choice(60, L, replace=False)] if L else [""]
    _PORT_CACHE[:] = []
    for s in (0, 100, 4321, N - 37):
        sl = arr[s:s + 128]
        a, b = pack_portfolios(sl, m), packed_portfolios_of(sl, m)
        W = a[0].shape[1]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0][:, :W]) and (b[0][:, W:] == -1).all()
    assert _PORT_CACHE and _PORT_CACHE[0][0] is arr
    arr[150] = ["000001", "000002", "000003", "000004", "000005", "000006", "000007", "000008", "000009"]
    a, b = pack_portfolios(arr[100:228], m), packed_portfolios_of(arr[100:228], m)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])                     # repacked directly
    small = arr[:10].copy()
    a, b = pack_portfolios(small, m), packed_portfolios_of(small, m)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a, b = pack_portfolios(list(arr[:5]), m), packed_portfolios_of(list(arr[:5]), m)    # a plain list of lists
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ------------------------------------------------------------------ round 6: the line's accounting helpers
def test_bench_families_are_single_kernels_and_profiles_are_pinned_to_the_round(tmp_path, monkeypatch):
    """bench.py: the two tile forms of the grouped weight gradients are families of their own (so trace_dominant's co-scheduled
    rate is the kernel's own work over the kernel's own resident time), the new attention kernels belong to their families, and
    the committed profile of a workload is the CURRENT round's file when it exists, the newest older one otherwise."""
    import bench
    eight = "void gemm_tn_group_bx_kernel<1, 8>(TnGroupDev)"
    four = "void gemm_tn_group_bx_kernel<1, 4>(TnGroupDev)"
    assert bench._family_has("gemm_tn_bx8", eight) and not bench._family_has("gemm_tn_bx8", four)
    assert bench._family_has("gemm_tn_bx", four) and not bench._family_has("gemm_tn_bx", eight)
    assert bench._family_has("attn_fwd", "void attn_fwd_ring_kernel<3, 2>(AttnDev)")
    assert bench._family_has("attn_fwd", "void attn_fwd_kernel<3, 4>(AttnDev)")
    assert bench._family_has("attn_bwd", "void attn_bwd_ring_kernel_direct<3, 2>(AttnDev)")
    assert not bench._family_has("attn_bwd", "void attn_bwd_runs_kernel<3, 2, false>(AttnDev)")
    assert bench.bx_products("gemm_tn_bx8") == 3
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    assert bench._summary_for("C2@512") is None
    (prof / "r4_summary_C2@512.json").write_text("{}")
    (prof / "r5_summary_C2@512.json").write_text("{}")
    assert bench._summary_for("C2@512").endswith("r5_summary_C2@512.json")          # no file of this round: the newest older one
    (prof / ("%s_summary_C2@512.json" % bench.PROFILE_ROUND)).write_text("{}")
    assert bench._summary_for("C2@512").endswith("%s_summary_C2@512.json" % bench.PROFILE_ROUND)
    assert bench._trace_for("C9@1") is None
    from pfotgnrec_amd import _lib
    assert len(_lib.PROF_KINDS) == 17 and _lib.PROF_KINDS[16] == "gemm_tn_bx8"


def test_fused_adam_ranges_cover_check_for_the_gradient_clear():
    """FusedAdam(zero_grads_in_step=True) may fold the gradient clear into its kernel only when its ranges tile the whole flat
    buffer: the coverage test used by step() on hand-made range lists."""
    covered = lambda lo, hi, total: bool(lo) and lo[0] == 0 and hi[-1] == total and all(hi[j] == lo[j + 1] for j in range(len(lo) - 1))
    assert covered([0], [100], 100)
    assert covered([0, 40, 60], [40, 60, 100], 100)             # split by step counts (the GRU tensors one step behind), still a tiling
    assert not covered([0, 60], [40, 100], 100)                 # a tensor without a gradient in between
    assert not covered([10], [100], 100) and not covered([0], [90], 100) and not covered([], [], 100)


def test_adjacent_row_blocks_and_step_count_keys():
    """functional.adjacent_rows: blocks that are a cut of one matrix come back as that matrix without a copy, anything else as
    None; optim._StepCounts: per-tensor step counts stored under id(parameter), reachable by tensor and by id."""
    import torch
    from pfotgnrec_amd.functional import adjacent_rows
    from pfotgnrec_amd.optim import _StepCounts
    m = torch.arange(60.0).reshape(10, 6)
    whole = adjacent_rows((m[0:2], m[2:4], m[4:10]))
    assert whole is not None and whole.data_ptr() == m.data_ptr() and torch.equal(whole, m)
    part = adjacent_rows((m[2:4], m[4:9]))
    assert part.data_ptr() == m[2:].data_ptr() and torch.equal(part, m[2:9])
    assert adjacent_rows((m[0:2], m[3:5])) is None                      # a gap
    assert adjacent_rows((m[2:4], m[0:2])) is None                      # out of order
    assert adjacent_rows((m[0:2], m[2:4].clone())) is None              # another buffer
    assert adjacent_rows((m[0:2], m[2:4, :3])) is None                  # another width / not contiguous
    assert adjacent_rows((m[0:2].reshape(2, 1, 6),)) is None            # not a matrix
    assert torch.equal(adjacent_rows((m[0:0], m[0:3])), m[0:3])         # an empty block (no negatives) is fine
    a, b = torch.zeros(3), torch.zeros(3)
    c = _StepCounts()
    c[a] = 4
    assert a in c and id(a) in c and b not in c and c[a] == 4 and c.get(b, 0) == 0 and c.get(id(a)) == 4
    d = c.copy()
    d[b] = 1
    assert isinstance(d, _StepCounts) and b not in c and d[b] == 1 and max(d.values()) == 4
