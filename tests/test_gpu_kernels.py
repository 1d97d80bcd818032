"""GPU parity, kernel by kernel, through the C ABI: integer/index work bit-exact, floating point within the
tolerance written at each assert.  Checked against the oracle and the reference-captured golden fixtures."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]

import pfotgnrec_amd as P
from pfotgnrec_amd import _lib
from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph, CONFIGS
from oracle.neighbor_finder import OracleNeighborFinder, build_adjacency
from oracle import mv_select as omv
from oracle import tgn_oracle as T

DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ------------------------------------------------------------------ K1
@pytest.mark.parametrize("K", [10, 3, 0])
def test_sampler_golden_recent(K):
    g = load_golden("g1_sampler")
    nf = P.NeighborFinder.from_arrays(g["a_src"], g["a_dst"], g["a_eidx"], g["a_ts"])
    nb, ei, et = nf.get_temporal_neighbor(g["a_q_nodes"], g["a_q_ts"], K)
    assert np.array_equal(nb, g["a_K%d_nbr" % K]) and np.array_equal(ei, g["a_K%d_eidx" % K])
    assert np.array_equal(et, g["a_K%d_et" % K]) and et.dtype == np.float32 and nb.dtype == np.int32


@pytest.mark.parametrize("K", [4, 20])
def test_sampler_golden_adversarial(K):
    """duplicate timestamps (strict <), zero-degree nodes, node 0, K > degree"""
    g = load_golden("g1_sampler")
    nf = P.NeighborFinder.from_arrays(g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"])
    nb, ei, et = nf.get_temporal_neighbor(g["b_q_nodes"], g["b_q_ts"], K)
    assert np.array_equal(nb, g["b_K%d_nbr" % K]) and np.array_equal(ei, g["b_K%d_eidx" % K])
    assert np.array_equal(et, g["b_K%d_et" % K])


def test_sampler_uniform_injected_draws():
    g = load_golden("g1_sampler")
    nf = P.NeighborFinder.from_arrays(g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"], uniform=True)
    nb, ei, et = nf.get_temporal_neighbor(g["b_q_nodes"], g["b_q_ts"], 5, draws=g["b_uni_draws"])
    onf = OracleNeighborFinder(*build_adjacency(g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"]), uniform=True)
    rn, re, rt = onf.gather_uniform(g["b_q_nodes"], g["b_q_ts"], g["b_uni_draws"], 5)
    assert np.array_equal(nb, rn) and np.array_equal(ei, re) and np.array_equal(et, rt)      # canonical stable order
    assert np.array_equal(et, g["b_uni_et"])                                                   # reference's sorted times
    for i in range(len(nb)):                                                                   # valid up to tie permutation
        for tt in np.unique(et[i]):
            m = et[i] == tt
            assert sorted(zip(nb[i][m], ei[i][m])) == sorted(zip(g["b_uni_nbr"][i][m], g["b_uni_eidx"][i][m]))


def test_sampler_uniform_philox_semantics():
    g = load_golden("g1_sampler")
    nf = P.NeighborFinder.from_arrays(g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"], uniform=True)
    onf = OracleNeighborFinder(*build_adjacency(g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"]))
    nb, ei, et = nf.get_temporal_neighbor(g["b_q_nodes"], g["b_q_ts"], 7)
    for i, (n_, t_) in enumerate(zip(g["b_q_nodes"], g["b_q_ts"])):
        hn, he, ht = onf.find_before(int(n_), t_)
        if len(hn) == 0:
            assert not nb[i].any() and not ei[i].any()
        else:
            hist = set(zip(hn.tolist(), he.tolist()))
            assert all((a, b) in hist for a, b in zip(nb[i].tolist(), ei[i].tolist()))
            assert np.all(np.diff(et[i]) >= 0)


def test_sampler_large_rows_and_frontier_expansion():
    """C2-like degrees (items with thousands of edges): 16-ary search vs searchsorted, bit-exact; plus the level expansion."""
    cfg = SyntheticConfig("big", 2000, 20, 60000, 8, 1, 20, 2)
    g = make_graph(cfg, with_prices=False, with_portfolios=False)
    d = g.data
    nf = P.get_neighbor_finder(d, False)
    onf = OracleNeighborFinder(*build_adjacency(d.sources, d.destinations, d.edge_idxs, d.timestamps))
    rs = np.random.RandomState(0)
    q = np.concatenate([d.destinations[30000:30400], d.sources[30000:30400], rs.randint(0, g.n_nodes, 200)])
    qt = np.concatenate([d.timestamps[30000:30400], d.timestamps[30000:30400], rs.randint(0, 1 << 24, 200).astype(np.float64)])
    nb, ei, et = nf.get_temporal_neighbor(q, qt, 20)
    rn, re, rt = onf.get_temporal_neighbor(q, qt, 20)
    assert np.array_equal(nb, rn) and np.array_equal(ei, re) and np.array_equal(et, rt)
    # raw ABI call with dt + frontier expansion
    indptr, anbr, aeidx, ats = nf.device_arrays(DEV)
    N, K = len(q), 20
    qn, qts = t(q.astype(np.int32)), t(qt)
    dt = torch.empty((N, K), dtype=torch.float32, device=DEV)
    nxt = torch.empty(N * (K + 1), dtype=torch.int32, device=DEV)
    nts = torch.empty(N * (K + 1), dtype=torch.float64, device=DEV)
    _lib.call("pfo_tnbr_sample", indptr.data_ptr(), anbr.data_ptr(), aeidx.data_ptr(), ats.data_ptr(), nf.n_nodes,
              qn.data_ptr(), qts.data_ptr(), N, K, 0, None, 0, 0, None, None, None, dt.data_ptr(), nxt.data_ptr(),
              nts.data_ptr(), _lib.stream_ptr())
    assert np.array_equal(dt.cpu().numpy(), (qt[:, None] - rt).astype(np.float32))          # embedding_module.py:133-135
    assert np.array_equal(nxt.cpu().numpy(), np.concatenate([q.astype(np.int32), rn.flatten()]))
    assert np.array_equal(nts.cpu().numpy(), np.concatenate([qt, np.repeat(qt, K)]))


# ------------------------------------------------------------------ candidate draw
@pytest.mark.parametrize("size", [3, 20, 30])
def test_candidate_draw_semantics(size):
    g = load_golden("g2_candidates")
    n_items, upper_u = int(g["n_items"]), int(g["upper_u"])
    codes = ["%06d" % (i + 1) for i in range(n_items)]
    m = {c: i for i, c in enumerate(codes)}
    ports = [[codes[j] for j in row[:n]] if n > 0 else [""] for row, n in zip(g["port_idx"], g["port_len"])]
    s = P.RandEdgeSampler(g["src"], g["dst_all"], ports, upper_u, m, seed=2024 if size == 30 else None)
    neg = s.sample(size)
    assert neg.shape == (len(ports), size) and neg.dtype == np.int64
    avail_all = set(np.unique(g["dst_all"]).tolist())
    for b in range(len(ports)):
        port = set((g["port_idx"][b][:g["port_len"][b]] + upper_u + 1).tolist())
        avail = avail_all - port
        assert set(neg[b].tolist()) <= avail                                   # utils.py:96
        if len(avail) >= size:
            assert len(set(neg[b].tolist())) == size                           # replace=False (utils.py:109-111)
    if size == 30:                                                             # seeded: reproducible (utils.py:82-84)
        assert np.array_equal(neg, P.RandEdgeSampler(g["src"], g["dst_all"], ports, upper_u, m, seed=2024).sample(size))
    else:
        assert not np.array_equal(neg, s.sample(size))
    # draws cover the available set roughly uniformly
    if size == 20:
        cnt = np.zeros(n_items)
        for _ in range(20):
            np.add.at(cnt, (s.sample(size) - upper_u - 1).flatten(), 1)
        assert cnt[list(i - upper_u - 1 for i in avail_all)].min() > 0


# ------------------------------------------------------------------ K2
@pytest.mark.parametrize("lam", [0.5, 0.1])
def test_mv_select_golden(lam):
    g = load_golden("g3_mv")
    pre = "lam%02d_" % int(lam * 10)
    upper_u = int(g["upper_u"])
    mvs = P.MVSampler(g["prices"], upper_u, DEV, gamma=float(g["gamma"]), lambda_mv=lam, p_pos_num=1, p_neg_num=3,
                      day_of=lambda ts: g[pre + "day_idx"])
    p_pos, p_neg, y, nr = mvs.select(g["dst"], g[pre + "neg"], g["ts"], g["port_idx"], g["port_len"], want_scores=True)
    assert np.allclose(y, g[pre + "y_mv"], rtol=1e-11, atol=0)                 # fp64; BLAS vs sequential summation order
    inv_rank = np.stack([omv.fuse_ranks(r, lam)[0] for r in y])
    assert np.array_equal(inv_rank, g[pre + "invest_rank"])                    # ranks are exact
    assert np.array_equal(nr, g[pre + "new_rank"])                             # lambda blend bit-exact
    cand = np.concatenate([g["dst"][:, None], g[pre + "neg"]], 1)
    n_tiefree = 0
    for b in range(len(cand)):
        order = omv.canonical_order(g[pre + "new_rank"][b])
        assert p_pos[b] == cand[b][order[0]] and np.array_equal(p_neg[3 * b:3 * b + 3], cand[b][order[-3:]])
        ref_order = g[pre + "order"][b]
        sel = list(ref_order[:1]) + list(ref_order[-3:])
        nrb = g[pre + "new_rank"][b]
        # tie-free selection -> identical to the reference's own (platform-sorted) output
        boundary_tie = (np.sum(nrb == nrb[ref_order[0]]) > 1) or any(np.sum(nrb == nrb[i]) > 1 for i in ref_order[-3:])
        if not boundary_tie:
            n_tiefree += 1
            assert p_pos[b] == g[pre + "p_pos"][b] and np.array_equal(p_neg[3 * b:3 * b + 3], g[pre + "p_neg"][3 * b:3 * b + 3])
    assert n_tiefree > 0


# ------------------------------------------------------------------ TimeEncode
def test_time_encode_golden():
    g = load_golden("g4_modules")
    y = P.time_encode(t(g["te_t"]), t(g["te_w"].reshape(-1)), t(g["te_b"])).cpu().numpy()
    assert np.abs(y - g["te_y"]).max() < 1e-6          # exact fp32 FMA argument (|t*w| up to 1.7e7) + ~1 ulp cosine


def test_time_encode_large_arguments_vs_float64():
    rs = np.random.RandomState(1)
    tt = np.concatenate([rs.randint(0, 1 << 24, 5000), rs.uniform(0, 3e10, 3000)]).astype(np.float32)
    w = (1 / 10 ** np.linspace(0, 9, 172)).astype(np.float32)
    b = rs.randn(172).astype(np.float32)
    y = P.time_encode(t(tt), t(w), t(b)).cpu().numpy()
    arg = T.fmaf(tt[:, None], w, b).astype(np.float64)
    assert np.abs(y - np.cos(arg)).max() < 5e-7


# ------------------------------------------------------------------ GEMM
def _gemm(A, B, bias, a_km, b_km, relu=0):
    M = A.shape[1] if a_km else A.shape[0]
    K = A.shape[0] if a_km else A.shape[1]
    N = B.shape[1] if b_km else B.shape[0]
    C = torch.full((M, N), float("nan"), device=DEV)
    ws = torch.empty(20_000_000, device=DEV)
    a, b = t(A), t(B)
    bb = t(bias) if bias is not None else None
    _lib.call("pfo_gemm_f32", a.data_ptr(), A.shape[1], int(a_km), b.data_ptr(), B.shape[1], int(b_km), C.data_ptr(), N,
              _lib.ptr(bb), M, N, K, relu, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    return C.cpu().numpy()


@pytest.mark.parametrize("M,N,K", [(1, 344, 172), (300, 344, 172), (257, 696, 172), (129, 172, 516), (64, 64, 32),
                                   (1000, 516, 520), (130, 86, 348), (77, 348, 86)])
def test_gemm_nt_and_nn(M, N, K):
    rs = np.random.RandomState(M + N + K)
    A = rs.randn(M, K).astype(np.float32)
    W = rs.randn(N, K).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    got = _gemm(A, W, bias, 0, 0)
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max() + 1e-5         # exact-fp32 MFMA, K <= 520
    got = _gemm(A, np.ascontiguousarray(W.T), None, 0, 1, relu=1)
    assert np.abs(got - np.maximum(ref - bias, 0)).max() < 2e-5 * np.abs(ref).max() + 1e-5


def _gemm_bf16x3(A, B, bias, b_km, relu=0):
    M, K = A.shape
    N = B.shape[1] if b_km else B.shape[0]
    C = torch.full((M, N), float("nan"), device=DEV)
    nbytes = _lib.load().pfo_gemm_bf16x3_workspace_bytes(N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    a, b = t(A), t(B)
    bb = t(bias) if bias is not None else None
    _lib.call("pfo_gemm_bf16x3", a.data_ptr(), K, b.data_ptr(), B.shape[1], int(b_km), C.data_ptr(), N, _lib.ptr(bb), M, N, K,
              relu, ws.data_ptr(), nbytes, _lib.stream_ptr())
    return C.cpu().numpy()


@pytest.mark.parametrize("M,N,K", [(1, 344, 172), (300, 348, 172), (257, 696, 172), (129, 172, 516), (64, 64, 32),
                                   (1000, 516, 520), (130, 86, 348), (77, 348, 88), (4099, 177, 44), (5000, 172, 696),
                                   (4500, 348, 172), (4300, 520, 88), (4097, 700, 36)])
def test_gemm_bf16x3_split_contraction_matches_float64(M, N, K):
    """fp32 contraction on the bf16 matrix cores (3-way operand split, 6 piece products): fp32-level accuracy.
    Operands with a wide dynamic range (exponents differ along k) so that dropped low pieces would show."""
    rs = np.random.RandomState(M + N + K)
    A = (rs.randn(M, K) * np.exp(2 * rs.randn(M, K))).astype(np.float32)
    W = (rs.randn(N, K) * np.exp(2 * rs.randn(N, K))).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    mag = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T + 1.0
    got = _gemm_bf16x3(A, W, bias, 0)
    assert (np.abs(got - (ref + bias)) / mag).max() < 4e-6            # an fp32 accumulation over K <= 696 gives ~2e-6
    got = _gemm_bf16x3(A, np.ascontiguousarray(W.T), None, 1, relu=1)  # k-major weights: image built transposed
    assert (np.abs(got - np.maximum(ref, 0)) / mag).max() < 4e-6
    # and it is no worse than the exact-fp32 MFMA kernel on the same operands
    f32 = _gemm(A, W, bias, 0, 0)
    bx = _gemm_bf16x3(A, W, bias, 0)
    assert (np.abs(bx - (ref + bias)) / mag).max() <= 2.0 * (np.abs(f32 - (ref + bias)) / mag).max() + 1e-7


@pytest.mark.parametrize("M", [300, 4200])              # the 32-row kernel / the 128-row kernel (pfo_gemm_bf16x3's own rule)
def test_gemm_split_contraction_row_scales(M):
    """The two-piece fp16 split scales every operand row by a power of two taken from the row's largest magnitude: activation
    rows from a RUNNING maximum over the k-tiles (the accumulators are rescaled when a later tile holds a larger value), weight
    rows from the image builder.  Rows whose magnitudes span the fp32 range, maxima that grow and shrink along k, rows of
    zeros and denormals: the result stays within the norm-wise bound 4e-6 * |a|.|b| row by row and column by column."""
    rs = np.random.RandomState(M)
    N, K = 348, 520
    A = rs.randn(M, K).astype(np.float64)
    W = rs.randn(N, K).astype(np.float64)
    A *= 2.0 ** rs.randint(-40, 40, size=(M, 1))            # every row its own magnitude, far outside fp16's range
    W *= 2.0 ** rs.randint(-20, 20, size=(N, 1))
    ramp = 2.0 ** np.linspace(-12, 12, K)                   # a maximum that grows with every k-tile ...
    A[0::7] *= ramp
    A[1::7] *= ramp[::-1]                                   # ... and one that shrinks
    A[2::7, K // 2:] = 0                                    # zeros behind a live half
    A[3::7] = 0                                             # rows of zeros
    if os.environ.get("PFO_BX_FMT") != "0":                 # (the bf16x3 path feeds denormal pieces to the matrix cores, which flush them)
        A[4::7] = 1e-42                                     # fp32 denormals
    W[5::11] = 0
    W[6::11] *= ramp[::-1]
    A = A.astype(np.float32); W = W.astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    # |a_i| . |b_n| with each factor's elements floored at 2^-16 of its row maximum: the bound of a per-row scale
    fa = np.maximum(np.abs(A).astype(np.float64), np.abs(A).max(1, keepdims=True).astype(np.float64) * 2.0 ** -16)
    fw = np.maximum(np.abs(W).astype(np.float64), np.abs(W).max(1, keepdims=True).astype(np.float64) * 2.0 ** -16)
    mag = fa @ fw.T
    got = _gemm_bf16x3(A, W, bias, 0).astype(np.float64)
    assert np.abs(ref).max() < 1e37 and np.isfinite(got).all()
    err = np.abs(got - (ref + bias)) / (mag + np.abs(bias) + 1e-30)
    assert err.max() < 4e-6, err.max()
    assert np.array_equal(got[3::7], np.broadcast_to(bias.astype(np.float64), got[3::7].shape))      # 0 . w + b = b exactly
    got_t = _gemm_bf16x3(A, np.ascontiguousarray(W.T), None, 1).astype(np.float64)                  # k-major weights
    assert (np.abs(got_t - ref) / (mag + 1e-30)).max() < 4e-6


@pytest.mark.parametrize("M,N,K", [(172, 172, 3000), (344, 348, 5001), (516, 520, 900), (86, 348, 2049), (344, 172, 40)])
def test_gemm_weight_gradient_form(M, N, K):
    """dW[M,N] = A[K,M]^T B[K,N] with split-K over workgroups and a deterministic slab reduce"""
    rs = np.random.RandomState(M + N + K)
    A = rs.randn(K, M).astype(np.float32)
    B = rs.randn(K, N).astype(np.float32)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    got = _gemm(A, B, None, 1, 1)
    assert np.abs(got - ref).max() < 3e-5 * np.abs(ref).max()
    # ... and PER ELEMENT against that element's own sum of magnitudes (Adam divides every gradient element by its own
    # sqrt(v): main.py:123,389 - a norm-wise bound would let small elements be all noise)
    mag = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
    assert (np.abs(got - ref) / mag).max() < 4e-6
    assert np.array_equal(got, _gemm(A, B, None, 1, 1))                       # bitwise reproducible


@pytest.mark.parametrize("grow", [True, False])
def test_gemm_weight_gradient_form_operand_scales(grow):
    """The weight-gradient tile's fp16 split keeps ONE power-of-two scale per operand and K-slab, a running maximum over the
    k-tiles: rows whose magnitude grows by 2^24 along k make the scale move (the accumulators are multiplied by the power of
    two in between) tile after tile; shrinking rows leave it where the first tile put it.  Columns a few binades apart, a
    block of zero rows.  Norm-wise bound against fp64, and bitwise reproducible."""
    rs = np.random.RandomState(7)
    M, N, K = 344, 348, 4096
    ramp = 2.0 ** np.linspace(-12, 12, K)
    if not grow:
        ramp = ramp[::-1]
    A = rs.randn(K, M) * ramp[:, None] * 2.0 ** rs.randint(-3, 4, size=(1, M))
    B = rs.randn(K, N) * ramp[::-1][:, None] * 2.0 ** rs.randint(-3, 4, size=(1, N))
    A[1000:1100] = 0
    A = A.astype(np.float32); B = B.astype(np.float32)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
    got = _gemm(A, B, None, 1, 1)
    assert np.abs(got - ref).max() < 4e-6 * mag.max()
    assert np.array_equal(got, _gemm(A, B, None, 1, 1))


def _tn_scale_floor(X, col_block, k_tile=32, floor=2.0 ** -14):
    """max(|x_kc|, floor * S(k, block(c))) with S = the largest magnitude of the operand's column block over every row up to
    the end of k's tile: an upper bound of the ONE running scale the weight-gradient tile keeps per operand, workgroup tile
    (128 columns of A, 176 of B: gemm.hip BM / BN) and K-slab (a slab starts later than row 0, so its running maximum is at
    most this one).  ``floor``: 2^-16 of the scaled top, which sits up to HX_GROW = 2 binades above the maximum."""
    K, C = X.shape
    out = np.abs(X).astype(np.float64)
    for c0 in range(0, C, col_block):
        blk = out[:, c0:c0 + col_block]
        per_tile = blk.reshape(-1, k_tile, blk.shape[1]).max((1, 2)) if K % k_tile == 0 else None
        assert per_tile is not None, "K must be a multiple of the k-tile in this test"
        run = np.maximum.accumulate(per_tile)                         # [tiles]
        out[:, c0:c0 + col_block] = np.maximum(blk, floor * np.repeat(run, k_tile)[:, None])
    return out


@pytest.mark.parametrize("shape", ["ramp_up", "ramp_down", "flat"])
def test_gemm_weight_gradient_form_per_element_bound(shape):
    """VERDICT r3 item 2b.  The weight-gradient tile's two-piece fp16 split keeps ONE power-of-two scale per operand, workgroup
    tile and K-slab, so an element far below its tile's largest magnitude loses RELATIVE precision; what the design claims is
      |dW_mn - exact| <= c * sum_k max(|a_km|, 2^-16 top_a) * max(|b_kn|, 2^-16 top_b)        PER ELEMENT (m, n),
    top = the (headroom-adjusted) running maximum of the operand's tile.  Operand columns 2^+-12 apart inside one tile, one
    column of A all tiny (2^-30 of its neighbours), magnitudes that grow / shrink by 2^24 along k: every element of the
    result is held to that bound - not the largest one only."""
    rs = np.random.RandomState(17)
    M, N, K = 344, 348, 4096
    ramp = {"ramp_up": 2.0 ** np.linspace(-12, 12, K), "ramp_down": 2.0 ** np.linspace(12, -12, K), "flat": np.ones(K)}[shape]
    A = rs.randn(K, M) * ramp[:, None] * 2.0 ** rs.randint(-12, 13, size=(1, M))
    B = rs.randn(K, N) * ramp[::-1][:, None] * 2.0 ** rs.randint(-12, 13, size=(1, N))
    A[:, 5] *= 2.0 ** -30                                            # a column that is tiny everywhere
    B[:, 200] *= 2.0 ** -30
    A[1000:1100] = 0
    A = A.astype(np.float32); B = B.astype(np.float32)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    bound = _tn_scale_floor(A, 128).T @ _tn_scale_floor(B, 176)
    got = _gemm(A, B, None, 1, 1).astype(np.float64)
    ratio = np.abs(got - ref) / bound
    assert ratio.max() < 4e-6, (ratio.max(), np.unravel_index(ratio.argmax(), ratio.shape))
    # columns that sit within 2^-12 of their tile's maximum are above the floor almost everywhere: for them the bound IS the
    # component-wise one, sum_k |a||b| - checked explicitly on the largest-magnitude columns of each operand
    big_a = np.abs(A).max(0) >= 2.0 ** -4 * np.abs(A).max()
    big_b = np.abs(B).max(0) >= 2.0 ** -4 * np.abs(B).max()
    mag = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
    if shape == "flat":
        assert (np.abs(got - ref) / mag)[np.ix_(big_a, big_b)].max() < 4e-6


# ------------------------------------------------------------------ BPR + Adam
def test_bpr_loss_and_gradient_vs_oracle():
    rs = np.random.RandomState(3)
    B, D, q = 37, 172, 3
    for pos_block in (1, 2):
        R = B * (pos_block + 1 + q)
        emb = (rs.randn(R, D) * 0.3).astype(np.float32)
        e = t(emb).requires_grad_(True)
        loss = P.bpr_loss(e, B, q, pos_block=pos_block)
        loss.backward()
        src, pos, neg = emb[:B], emb[pos_block * B:(pos_block + 1) * B], emb[(pos_block + 1) * B:]
        rl, cache = T.bpr_loss(src, pos.reshape(B, 1, D), neg.reshape(B, q, D))
        ds, dp, dn = T.bpr_loss_backward(cache)
        ref = np.zeros_like(emb)
        ref[:B] = ds; ref[pos_block * B:(pos_block + 1) * B] += dp.reshape(B, D); ref[(pos_block + 1) * B:] = dn.reshape(B * q, D)
        assert abs(float(loss) - float(rl)) < 1e-5
        assert np.abs(e.grad.cpu().numpy() - ref).max() < 1e-6 * max(1.0, np.abs(ref).max()) + 1e-7


def test_fused_adam_matches_torch_adam():
    rs = np.random.RandomState(0)
    n = 100_003
    p0 = rs.randn(n).astype(np.float32)
    ref_p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([ref_p], lr=1e-3)
    p = t(p0.copy()); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for step in range(1, 6):
        g = (rs.randn(n) * (10.0 ** rs.randint(-6, 2))).astype(np.float32)
        ref_p.grad = torch.from_numpy(g.copy())
        opt.step()
        gd = t(g)
        _lib.call("pfo_adam_step", p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, step,
                  _lib.stream_ptr())
    assert np.abs(p.cpu().numpy() - ref_p.detach().numpy()).max() < 2e-6


def test_device_csr_build_equals_host_build():
    g = load_golden("g1_sampler")
    from pfotgnrec_amd.neighbor_finder import build_csr
    for p in ("a", "b"):                                   # b has heavy timestamp ties: the sorts must be stable
        host = build_csr(g[p + "_src"], g[p + "_dst"], g[p + "_eidx"], g[p + "_ts"])
        nf = P.NeighborFinder.from_arrays(g[p + "_src"], g[p + "_dst"], g[p + "_eidx"], g[p + "_ts"], device=DEV)
        for a, b in zip(host, (nf.indptr, nf.nbr, nf.eidx, nf.ts)):
            assert np.array_equal(a, b)
    nb, ei, et = nf.get_temporal_neighbor(g["b_q_nodes"], g["b_q_ts"], 4)
    assert np.array_equal(nb, g["b_K4_nbr"]) and np.array_equal(ei, g["b_K4_eidx"])


# ------------------------------------------------------------------ f-2: device CSR build (native radix sort) + append
def _csr_equal(a, b):
    return all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))


@pytest.mark.parametrize("which", ["a", "b", "shuffled", "c2"])
def test_native_csr_build_equals_host_build(which):
    """pfo_csr_build (stable LSD radix sort on the device) against the host lexsort build, which g1 pins to the reference's
    NeighborFinder: the g1 graphs (adversarial timestamp ties, zero-degree nodes), a log whose timestamps are NOT
    chronological (the eight timestamp passes run), and the 1 M-edge C2 graph."""
    from pfotgnrec_amd.neighbor_finder import build_csr, build_csr_device
    if which in ("a", "b"):
        g = load_golden("g1_sampler")
        src, dst, eidx, ts = g[which + "_src"], g[which + "_dst"], g[which + "_eidx"], g[which + "_ts"]
    elif which == "shuffled":
        rs = np.random.RandomState(3)
        E = 30000
        src, dst = rs.randint(1, 400, E), rs.randint(400, 460, E)
        ts = rs.randint(0, 500, E).astype(np.float64) - 100.0                     # unsorted, heavy ties, negative values
        ts[::7] += 0.25
        eidx = rs.permutation(E) + 1
    else:
        gr = make_graph(CONFIGS["C2"], with_prices=False, with_portfolios=False)
        src, dst, eidx, ts = gr.data.sources, gr.data.destinations, gr.data.edge_idxs, gr.data.timestamps
    host = build_csr(src, dst, eidx, ts, max_node_idx=int(max(src.max(), dst.max())) + 3)
    devc = build_csr_device(src, dst, eidx, ts, DEV, max_node_idx=int(max(src.max(), dst.max())) + 3)
    assert _csr_equal(host, [x.cpu().numpy() for x in devc])


def test_neighbor_finder_append_equals_rebuild_and_reference():
    """NeighborFinder.append: the adversarial g1 graph built from its first third and grown twice equals the one-shot build
    (tie order included) and answers the fixture's queries exactly like the reference; node ids beyond the table grow it."""
    g = load_golden("g1_sampler")
    src, dst, eidx, ts = g["b_src"], g["b_dst"], g["b_eidx"], g["b_ts"]
    E = len(src)
    cuts = [0, E // 3, E // 2, E]
    nf = P.NeighborFinder.from_arrays(src[:cuts[1]], dst[:cuts[1]], eidx[:cuts[1]], ts[:cuts[1]], device=DEV)
    for a, b in zip(cuts[1:-1], cuts[2:]):
        nf.append(src[a:b], dst[a:b], eidx[a:b], ts[a:b], device=DEV)
    full = P.NeighborFinder.from_arrays(src, dst, eidx, ts)
    assert nf.n_nodes == full.n_nodes
    assert _csr_equal((nf.indptr, nf.nbr, nf.eidx, nf.ts), (full.indptr, full.nbr, full.eidx, full.ts))
    for K in (4, 20):
        nb, ei, et = nf.get_temporal_neighbor(g["b_q_nodes"], g["b_q_ts"], K)
        assert np.array_equal(nb, g["b_K%d_nbr" % K]) and np.array_equal(ei, g["b_K%d_eidx" % K]) and np.array_equal(et, g["b_K%d_et" % K])
    # out-of-order arrival: appending edges OLDER than what is there lands them in timestamp order, behind equal timestamps
    rs = np.random.RandomState(9)
    m = 50
    s2, d2 = rs.randint(1, 10, m), rs.randint(13, 19 + 4, m)                   # ids 19..22 are new nodes
    t2 = rs.randint(0, 25, m).astype(np.float64)
    e2 = np.arange(E + 1, E + m + 1)
    nf.append(s2, d2, e2, t2, device=DEV)
    both = P.NeighborFinder.from_arrays(np.concatenate([src, s2]), np.concatenate([dst, d2]), np.concatenate([eidx, e2]),
                                        np.concatenate([ts, t2]))
    assert nf.n_nodes == both.n_nodes == 23
    assert _csr_equal((nf.indptr, nf.nbr, nf.eidx, nf.ts), (both.indptr, both.nbr, both.eidx, both.ts))
