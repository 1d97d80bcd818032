"""Pin the oracle (oracle/*.py) against fixtures captured from the reference itself (tools/make_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.neighbor_finder import OracleNeighborFinder, build_adjacency
from oracle.rand_edge_sampler import OracleRandEdgeSampler
from oracle import mv_select as mv


# ---------------------------------------------------------------- G1 sampler (bit-exact)
@pytest.mark.parametrize("K", [10, 3, 0])
def test_g1_most_recent(K):
    g = load_golden("g1_sampler")
    nf = OracleNeighborFinder(*build_adjacency(g["a_src"], g["a_dst"], g["a_eidx"], g["a_ts"]))
    nb, ei, et = nf.get_temporal_neighbor(g["a_q_nodes"], g["a_q_ts"], K)
    for got, key in ((nb, "nbr"), (ei, "eidx"), (et, "et")):
        ref = g["a_K%d_%s" % (K, key)]
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("K", [4, 20])
def test_g1_adversarial(K):
    g = load_golden("g1_sampler")
    nf = OracleNeighborFinder(*build_adjacency(g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"]))
    nb, ei, et = nf.get_temporal_neighbor(g["b_q_nodes"], g["b_q_ts"], K)
    assert np.array_equal(nb, g["b_K%d_nbr" % K])
    assert np.array_equal(ei, g["b_K%d_eidx" % K])
    assert np.array_equal(et, g["b_K%d_et" % K])


def test_g1_uniform_same_rng_stream():
    """Same numpy calls in the same order: seeding the global RNG reproduces the reference bit for bit."""
    g = load_golden("g1_sampler")
    nf = OracleNeighborFinder(*build_adjacency(g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"]), uniform=True)
    np.random.seed(int(g["b_uni_seed"]))
    log = []
    nb, ei, et = nf.get_temporal_neighbor(g["b_q_nodes"], g["b_q_ts"], 5, draw_log=log)
    assert np.array_equal(nb, g["b_uni_nbr"]) and np.array_equal(ei, g["b_uni_eidx"]) and np.array_equal(et, g["b_uni_et"])
    draws = g["b_uni_draws"]
    for i, _, idx in log:
        assert np.array_equal(idx, draws[i])


def test_g1_uniform_injected_draws_canonical_sort():
    """Injected draws + stable re-sort: equal to the reference up to permutations inside equal-time groups (App. A-9)."""
    g = load_golden("g1_sampler")
    nf = OracleNeighborFinder(*build_adjacency(g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"]), uniform=True)
    nb, ei, et = nf.gather_uniform(g["b_q_nodes"], g["b_q_ts"], g["b_uni_draws"], 5)
    assert np.array_equal(et, g["b_uni_et"])                      # times are sorted either way
    for i in range(len(nb)):
        for t in np.unique(et[i]):
            m = et[i] == t
            assert sorted(zip(nb[i][m], ei[i][m])) == sorted(zip(g["b_uni_nbr"][i][m], g["b_uni_eidx"][i][m]))


# ---------------------------------------------------------------- G2 candidate draw
def _portfolios(g, codes):
    return [[codes[j] for j in row[:n]] if n > 0 else [""] for row, n in zip(g["port_idx"], g["port_len"])]


@pytest.mark.parametrize("size", [3, 20, 30])
def test_g2_candidates(size):
    g = load_golden("g2_candidates")
    n_items, upper_u = int(g["n_items"]), int(g["upper_u"])
    codes = ["%06d" % (i + 1) for i in range(n_items)]
    map_item_id = {c: i for i, c in enumerate(codes)}
    seed = int(g["seed_size%d" % size])
    np.random.seed(5)
    avail = []
    s = OracleRandEdgeSampler(g["src"], g["dst_all"], _portfolios(g, codes), upper_u, map_item_id, seed=None if seed < 0 else seed)
    neg = s.sample(size, available_log=avail)
    ref = g["neg_size%d" % size]
    assert np.array_equal(neg, ref)                               # same RNG stream, same calls
    for b in range(len(ref)):                                      # semantics the device draw must keep (App. A-8)
        assert set(ref[b]) <= set(avail[b])
        port = set(g["port_idx"][b][:g["port_len"][b]] + upper_u + 1)
        assert not (set(ref[b]) & port)
        if len(avail[b]) >= size:
            assert len(set(ref[b])) == size


# ---------------------------------------------------------------- G3 MV selection
@pytest.mark.parametrize("lam", [0.5, 0.1])
def test_g3_mv(lam):
    g = load_golden("g3_mv")
    pre = "lam%02d_" % int(lam * 10)
    upper_u = int(g["upper_u"])
    neg = g[pre + "neg"]
    cand = np.concatenate([g["dst"][:, None], neg], 1) - (upper_u + 1)
    p_pos, p_neg, Y, NR = mv.mv_select(g["prices"], g[pre + "day_idx"], cand, g["port_idx"], g["port_len"],
                                       float(g["gamma"]), lam, 1, 3, platform_order=True)
    assert np.array_equal(Y, g[pre + "y_mv"])                      # same numpy calls -> bit-exact fp64
    assert np.array_equal(NR, g[pre + "new_rank"])
    assert np.array_equal(p_pos.flatten() + upper_u + 1, g[pre + "p_pos"])
    assert np.array_equal(p_neg.flatten() + upper_u + 1, g[pre + "p_neg"])
    # canonical tie policy: identical wherever the selection is tie-free, a valid tie permutation elsewhere
    cp, cn, _, _ = mv.mv_select(g["prices"], g[pre + "day_idx"], cand, g["port_idx"], g["port_len"], float(g["gamma"]), lam, 1, 3)
    n_tiefree = 0
    for b in range(len(cand)):
        nr = NR[b]
        order_ref = g[pre + "order"][b]
        order_can = mv.canonical_order(nr)
        assert np.array_equal(nr[order_ref], nr[order_can])        # same rank sequence
        uniq = len(np.unique(nr)) == len(nr)
        if uniq:
            n_tiefree += 1
            assert np.array_equal(cp[b], p_pos[b]) and np.array_equal(cn[b], p_neg[b])
    assert n_tiefree >= 0


def test_g3_both_branches_present():
    g = load_golden("g3_mv")
    assert (g["port_len"] == 0).any() and (g["port_len"] > 0).any()
