"""Round-4 GPU tests: the bench's rank path on RCCL at world 1, (later sections) oracle parity at the benched dropout and
per-element evidence for the fp16-split contractions."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]

if has_gpu():
    import pfotgnrec_amd as P
    from pfotgnrec_amd import _lib
    DEV = torch.device("cuda:0")


# ------------------------------------------------------------------ bench.py as a rank
@pytest.mark.parametrize("allreduce", ["single", "buckets", "fused", "fused_buckets"])
def test_bench_rank_path_on_rccl_world1(allreduce):
    """`bench.py --gpus 1` under a torchrun-style environment (RANK=0, WORLD_SIZE=1, MASTER_*) with PFO_DIST_FORCE=1: the
    process group is initialised on RCCL (device bound, explicit timeout), every step all-reduces the real flat gradient
    buffer through ncclAllReduce (one piece / two buckets with the backward's event), the timed blocks are bracketed by
    barriers and a MAX all-reduce, and the line carries collective_ms_per_step / compute_ms_per_step and the secondary run
    with the other all-reduce form.  A child process with a hard limit: a hang cannot take the suite down."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29300 + (os.getpid() % 200) + {"single": 0, "buckets": 7, "fused": 13, "fused_buckets": 19}[allreduce]), PFO_DIST_FORCE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("PFO_DIST_BACKEND", None)
    out = subprocess.run(["timeout", "-k", "10", "420", sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--config", "C2",
                          "--batch", "64", "--steps", "8", "--warmup", "3", "--min-seconds", "0.2", "--no-cpu-baseline", "--graph", "off",
                          "--allreduce", allreduce], env=env, capture_output=True)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    rec = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])
    cfg = rec["config"]
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert cfg["collective"].startswith("rccl all-reduce"), cfg["collective"]
    assert cfg["allreduce"] == allreduce
    assert cfg["collective_samples"] >= 1 and 0 < cfg["collective_ms_per_step"] < rec["ms_per_step"]
    if not allreduce.startswith("fused"):       # (fused: ncclAllReduce runs on the library's side stream inside bpr_step, beside the next batch's head)
        assert cfg["collective_on_callers_stream"]
        assert abs(cfg["compute_ms_per_step"] + cfg["collective_ms_per_step"] - rec["ms_per_step"]) < 1e-3
    else:
        assert not cfg["collective_on_callers_stream"] and abs(cfg["compute_ms_per_step"] - rec["ms_per_step"]) < 1e-3
    sec = rec["secondary"]
    assert "error" not in sec, sec
    for other in ("single", "buckets", "fused", "fused_buckets"):
        if other != allreduce:
            assert sec["allreduce_" + other]["value"] > 0
            assert sec["allreduce_" + other]["collective_ms_per_step"] > 0


# ------------------------------------------------------------------ oracle parity at the benched dropout
from test_gpu_tgn_step import _masked_bpr_backward, _legal_draws, relerr  # noqa: E402

RTOL_EMB = 1e-4            # north_star: embeddings within 1e-4 relative
RTOL_GRAD_L2 = 5e-3        # small shapes: relative L2 with near-kink roots left out on both sides (test_gpu_tgn_step.py)
RTOL_GRAD_TIME = 3e-3


def _oracle_for(tgn, g, d, L, H, use_mem, uniform):
    from oracle import tgn_oracle as T
    from oracle.neighbor_finder import OracleNeighborFinder, build_adjacency
    onf = OracleNeighborFinder(*build_adjacency(d.sources, d.destinations, d.edge_idxs, d.timestamps), uniform=uniform)
    names = [k for k in tgn.state_dict() if "layer_norm" not in k and not k.startswith("memory.")]
    ref = T.OracleTGN(onf, g.node_features, g.edge_features, {k: tgn.state_dict()[k].cpu().numpy() for k in names}, L, H, use_mem)
    return onf, names, ref


def test_dropout_mask_export_statistics_and_determinism():
    """pfo_attn_dropout_mask: multipliers are 0 or 1/(1-p), the drop rate is p (binomial bound), masks differ between layers,
    heads and steps and are reproduced exactly by a second export."""
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    cfg = SyntheticConfig("dm", 200, 20, 3000, 32, 2, 10, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.25,
                use_memory=True, memory_dimension=32, message_function="identity", n_neighbors=10)
    tgn.train()
    B = 64
    s = 1500
    neg = np.random.RandomState(0).randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)
    tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B], d.edge_idxs[s:s + B], 10)
    m1 = tgn.debug_dropout_masks()
    m1b = tgn.debug_dropout_masks()
    assert m1[2].shape == (5 * B, 2, 10) and m1[1].shape == (5 * B * 11, 2, 10)
    for l in (1, 2):
        assert np.array_equal(m1[l], m1b[l])
        vals = np.unique(m1[l])
        assert len(vals) == 2 and vals[0] == 0 and abs(vals[1] - 1 / 0.75) < 1e-6
        n = m1[l].size
        assert abs((m1[l] == 0).mean() - 0.25) < 5 * np.sqrt(0.25 * 0.75 / n)
        assert not np.array_equal(m1[l][:, 0], m1[l][:, 1])                     # heads draw independently
    assert not np.array_equal(m1[1][:5 * B], m1[2])                             # layers draw independently
    tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B], d.edge_idxs[s:s + B], 10)
    assert not np.array_equal(tgn.debug_dropout_masks()[2], m1[2])              # a new step, a new position in the stream
    tgn.eval()
    with torch.no_grad():
        tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B], d.edge_idxs[s:s + B], 10)
    assert tgn.debug_dropout_masks() is None


@pytest.mark.parametrize("D,H,L,K,use_mem,uniform,pdrop", [(32, 2, 1, 10, True, False, 0.1), (172, 2, 2, 8, True, False, 0.1),
                                                            (172, 4, 2, 6, False, True, 0.1), (64, 1, 2, 5, True, False, 0.5),
                                                            (128, 2, 2, 20, True, False, 0.3),   # K = 20: run-merged layer-1 backward
                                                            (256, 4, 1, 12, True, False, 0.1)])
def test_step_with_dropout_against_oracle(D, H, L, K, use_mem, uniform, pdrop):
    """Training steps in TRAIN mode with attention dropout (the setting bench.py times; temporal_attention.py:28,70) against
    the oracle replaying the step with the SAME masks (exported through the C ABI; the oracle's dropout algebra is pinned to
    the reference by the g8 fixture).  Embeddings 1e-4, loss, kink-masked gradients, the memory state machine; four steps
    with Adam in between, the oracle re-injected with the product's parameters every step."""
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    torch.manual_seed(4321 + D + H)
    cfg = SyntheticConfig("t", 300, 25, 5000, D, L, K, H)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    nf = P.get_neighbor_finder(d, uniform=uniform)
    tgn = P.TGN(nf, g.node_features, g.edge_features, DEV, n_layers=L, n_heads=H, dropout=pdrop, use_memory=use_mem,
                memory_dimension=D, message_function="identity", n_neighbors=K)
    with torch.no_grad():
        tgn.time_encoder.w.bias.normal_(0, 0.3)
        for att in tgn.embedding_module.attention_models:
            att.multi_head_target.in_proj_bias.normal_(0, 0.1)
            att.multi_head_target.out_proj.bias.normal_(0, 0.1)
    opt = P.FusedAdam(tgn, lr=1e-3)
    onf, names, ref = _oracle_for(tgn, g, d, L, H, use_mem, uniform)
    rs = np.random.RandomState(6)
    B = 40
    for step in range(4):
        s = 2500 + step * B
        sb, db, tb, eb = d.sources[s:s + B], d.destinations[s:s + B], d.timestamps[s:s + B], d.edge_idxs[s:s + B]
        neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)
        ref.P = {k: tgn.state_dict()[k].detach().cpu().numpy().copy() for k in names}
        draws, odraws = None, None
        if uniform:
            R = 5 * B
            raw = [rs.randint(0, 1 << 30, size=(R * (1 + K) ** i, K)).astype(np.int64) for i in range(L)]
            draws, odraws = _legal_draws(onf, np.concatenate([sb, db, neg]), np.concatenate([tb, tb, np.repeat(tb, 3)]), K, L, raw)
        tgn.train(); opt.zero_grad()
        se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, K, draws=draws)
        ref.dropout_masks = tgn.debug_dropout_masks()
        assert ref.dropout_masks is not None and (ref.dropout_masks[L] == 0).any()
        rse, rde, rne = ref.compute_temporal_embeddings(sb, db, neg, tb, eb, K, draws=None if odraws is None else list(odraws))
        emb = torch.cat([se, de, ne])
        remb = np.concatenate([rse, rde, rne])
        e = relerr(emb.detach().cpu().numpy(), remb)
        assert e < RTOL_EMB, (step, e)
        rgrads = _masked_bpr_backward(tgn, ref, emb, rse, rde, rne, B, K)
        for name, p in tgn.named_parameters():
            if name not in rgrads:
                continue
            r = rgrads[name].reshape(p.shape)
            if np.abs(r).max() < 1e-7:
                assert p.grad is None or p.grad.abs().max().item() < 1e-6, name
                continue
            got = p.grad.cpu().numpy().astype(np.float64)
            err = np.linalg.norm(got - r) / (np.linalg.norm(r) + 1e-30)
            assert err < (RTOL_GRAD_TIME if name.startswith("time_encoder") else RTOL_GRAD_L2), (step, name, err)
        if use_mem:
            assert relerr(tgn.memory.memory.cpu().numpy(), ref.memory) < RTOL_EMB
            assert np.array_equal(tgn.memory.last_update.cpu().numpy(), ref.last_update)
            tab, mt, has = ref.pending_table()
            assert np.array_equal(tgn.memory.has_msg.cpu().numpy() > 0, has)
            assert relerr(tgn.memory.msg_table.cpu().numpy()[has], tab[has]) < RTOL_EMB
        opt.step()
    # the masks matter: the same step replayed WITHOUT them is far outside the bar (the test would notice a no-op dropout)
    ref.dropout_masks = None
    ref.P = {k: tgn.state_dict()[k].detach().cpu().numpy().copy() for k in names}
    tgn.train()
    se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, K, draws=draws)
    rse, _, _ = ref.compute_temporal_embeddings(sb, db, neg, tb, eb, K, draws=None if odraws is None else list(odraws))
    assert relerr(se.detach().cpu().numpy(), rse) > 10 * RTOL_EMB


def test_step_against_reference_golden_at_dropout_01():
    """g8: one training step of the REFERENCE itself in train mode at dropout 0.1 (main.py's default), with the multipliers
    its own F.dropout applied captured per attention call.  The product takes them as injected decisions
    (pfo_tgn_batch.dropout_keep, the dropout counterpart of the uniform sampler's injected draws) and must land on the
    reference's embeddings (1e-4), loss, every parameter gradient (5e-4; time encoder 3e-3) and memory state."""
    from conftest import load_golden
    from test_gpu_tgn_step import inject
    g = load_golden("g8_dropout")
    L, H, K = int(g["step_L"]), int(g["step_H"]), int(g["step_K"])
    nf = P.NeighborFinder.from_arrays(g["src_all"], g["dst_all"], g["eidx_all"], g["ts_all"], uniform=False)
    D = g["node_features"].shape[1]
    tgn = P.TGN(nf, g["node_features"], g["edge_features"], DEV, n_layers=L, n_heads=H, dropout=float(g["step_p"]), use_memory=True,
                memory_dimension=D, message_function="identity", n_neighbors=K)
    inject(tgn, g, "s_")
    sb, db, tb, eb, neg = g["s_src"], g["s_dst"], g["s_ts"], g["s_eidx"], g["s_neg"]
    B = len(sb)
    masks = {1: g["s_drop_l1"], 2: g["s_drop_l2"]}
    tgn.train()
    se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg.flatten(), tb, eb, K, dropout_keep=masks)
    for got, key in ((se, "emb_src"), (de, "emb_dst"), (ne, "emb_neg")):
        e = relerr(got.detach().cpu().numpy(), g["s_" + key])
        assert e < RTOL_EMB, (key, e)
    emb = torch.cat([se, de, ne])
    loss = P.bpr_loss(emb, B, 3)
    assert abs(float(loss) - float(g["s_loss"])) < 1e-5 * max(1.0, abs(float(g["s_loss"])))
    loss.backward()
    n_checked = 0
    params = dict(tgn.named_parameters())
    for k in g.files:
        if not k.startswith("s_grad_"):
            continue
        name = k[len("s_grad_"):]
        if "layer_norm" in name or name.startswith("memory."):
            continue
        ref = g[k]
        got = params[name].grad.cpu().numpy()
        if np.abs(ref).max() < 1e-7:
            assert np.abs(got).max() < 1e-6, name
            continue
        e = relerr(got, ref)
        assert e < (3e-3 if name.startswith("time_encoder") else 5e-4), (name, e)
        n_checked += 1
    assert n_checked >= 20
    assert relerr(tgn.memory.memory.cpu().numpy(), g["s_after_memory"]) < RTOL_EMB
    assert np.array_equal(tgn.memory.last_update.cpu().numpy(), g["s_after_last_update"])
    has = g["s_after_msg_cnt"] > 0
    assert np.array_equal(tgn.memory.has_msg.cpu().numpy() > 0, has)
    assert relerr(tgn.memory.msg_table.cpu().numpy()[has], g["s_after_msg_tab"][has]) < RTOL_EMB
    # the injected decisions are what ran: without them (the step's own Philox masks) the embeddings are far off
    inject(tgn, g, "s_")
    with torch.no_grad():
        m = tgn.memory
        m.memory.copy_(torch.from_numpy(g["s_sd_memory.memory"])); m.last_update.copy_(torch.from_numpy(g["s_sd_memory.last_update"]))
    se2, _, _ = tgn.compute_temporal_embeddings(sb, db, neg.flatten(), tb, eb, K)
    assert relerr(se2.detach().cpu().numpy(), g["s_emb_src"]) > 10 * RTOL_EMB


# ------------------------------------------------------------------ parameter cache
@pytest.mark.parametrize("opt_kind", ["fused", "torch"])
def test_parameter_cache_changes_when_the_composites_are_built_not_what_they_are(opt_kind):
    """Composite weights and weight images depend on the parameters only (the reference re-reads nn.Linear weights and
    recomputes nothing per batch, tgn.py:219-327).  With the parameter cache they are built once per parameter version - by
    the first forward, by ``pfo_tgn_refresh`` right behind FusedAdam's kernel, or by the next forward after a torch optimizer
    / load_state_dict bumped the version counter - instead of by every forward.  Six training steps, an evaluation pass in
    between and a state_dict reload: embeddings, gradients (deterministic mode: bitwise), parameters and memory are
    IDENTICAL to the per-step rebuild."""
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    cfg = SyntheticConfig("pc", 300, 25, 6000, 64, 2, 8, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    B, q, K = 64, 3, 8
    rs = np.random.RandomState(8)
    starts = [3000 + B * i for i in range(6)]
    negs = [rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * q) for _ in starts]

    def run(cache, refresh):
        torch.manual_seed(21)
        tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.1,
                    use_memory=True, memory_dimension=64, message_function="identity", n_neighbors=K)
        tgn.param_cache, tgn.refresh_after_step = cache, refresh
        tgn.deterministic = True
        opt = P.FusedAdam(tgn, lr=1e-3) if opt_kind == "fused" else torch.optim.Adam(tgn.parameters(), lr=1e-3)
        out = []
        saved = None
        for i, (s, neg) in enumerate(zip(starts, negs)):
            tgn.train()
            se, de, ne = tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B],
                                                         d.edge_idxs[s:s + B], K)
            emb = torch.cat([se, de, ne])
            P.bpr_loss(emb, B, q).backward()
            grad = tgn.flat_grad.clone()
            opt.step()
            opt.zero_grad(set_to_none=True)
            out.append((emb.detach().clone(), grad, tgn.flat_parameters.detach().clone()))
            if i == 1:
                saved = {k: v.clone() for k, v in tgn.state_dict().items()}
            if i == 2:                                   # an evaluation pass (forward only, several calls on one parameter version)
                bk = tgn.memory.backup_memory()
                tgn.eval()
                with torch.no_grad():
                    for _ in range(2):
                        ev = torch.cat(tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg,
                                                                       d.timestamps[s:s + B], d.edge_idxs[s:s + B], K))
                out.append((ev.clone(), None, None))
                tgn.memory.restore_memory(bk)
            if i == 3:                                   # parameters replaced from outside (version counter bump)
                tgn.load_state_dict(saved)
        torch.cuda.synchronize()
        return out, tgn.memory.memory.detach().clone()

    base, mem0 = run(False, False)
    for cache, refresh in ((True, True), (True, False)):
        got, mem1 = run(cache, refresh)
        for (e0, g0, p0), (e1, g1, p1) in zip(base, got):
            assert torch.equal(e0, e1)
            assert g0 is None or (torch.equal(g0, g1) and torch.equal(p0, p1))
        assert torch.equal(mem0, mem1)


def test_backward_after_a_parameter_update_is_refused():
    """The backward reads the forward's composites from the parameter cache: an optimizer step between a forward and its
    backward (torch raises 'modified by an inplace operation' for the same mistake) is an error, not a silent mix."""
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    cfg = SyntheticConfig("pc2", 200, 20, 3000, 32, 2, 6, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.0,
                use_memory=True, memory_dimension=32, message_function="identity", n_neighbors=6)
    B, s = 32, 1500
    neg = np.random.RandomState(0).randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)
    tgn.train()
    emb = torch.cat(tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B], d.edge_idxs[s:s + B], 6))
    with torch.no_grad():
        tgn.embedding_module.attention_models[0].merger.fc1.weight.mul_(1.5)
    with pytest.raises(RuntimeError, match="modified between"):
        P.bpr_loss(emb, B, 3).backward()


def test_gru_gate_backward_as_gemm_epilogue_equals_the_separate_kernel():
    """PFO_FUSE_GATES=1: the GRU's gate backward runs as the epilogue of layer 1's dx_tab contraction (32-row image kernel)
    instead of its own launch.  Same arithmetic per element on the same inputs: gradients of a step are identical up to the
    summation order of nothing - bitwise in deterministic mode... the fused form is taken only for the float table (one
    replica, not the deterministic int64 one), so the comparison is at fp32 rounding of the atomics: 1e-5 relative."""
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    cfg = SyntheticConfig("fg", 300, 25, 6000, 64, 2, 8, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    B, K = 64, 8
    s = 3000
    neg = np.random.RandomState(3).randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)

    def grads(fuse):
        os.environ["PFO_FUSE_GATES"] = "1" if fuse else "0"
        try:
            torch.manual_seed(5)
            tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.0,
                        use_memory=True, memory_dimension=64, message_function="identity", n_neighbors=K)
            tgn.train()
            for s0 in (s - 2 * B, s - B):                     # populate memory and pending messages
                with torch.no_grad():
                    tgn.compute_temporal_embeddings(d.sources[s0:s0 + B], d.destinations[s0:s0 + B], neg, d.timestamps[s0:s0 + B],
                                                    d.edge_idxs[s0:s0 + B], K)
            emb = torch.cat(tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B],
                                                            d.edge_idxs[s:s + B], K))
            P.bpr_loss(emb, B, 3).backward()
            torch.cuda.synchronize()
            return tgn.flat_grad.clone()
        finally:
            os.environ.pop("PFO_FUSE_GATES", None)
    g0, g1 = grads(False), grads(True)
    gru = slice(2 * 64, 2 * 64 + 3 * 64 * (3 * 64 + 4) + 3 * 64 * 64 + 6 * 64)      # the GRU block of the flat layout
    assert g0[gru].abs().max().item() > 0
    assert (g0 - g1).abs().max().item() <= 1e-5 * g0.abs().max().item()


@pytest.mark.parametrize("use_memory", [True, False])
def test_fused_backward_and_optimizer_step_on_the_side_stream_equals_the_serial_order(use_memory):
    """``bpr_step(..., optimizer=opt)``: the end of the backward and the Adam kernel stay on the library's side stream while
    the caller's stream goes on to the next batch's sampling (pfo_tgn_batch.defer_join, pfo_tgn_adam_side); the next forward
    joins behind its neighbour sampling.  It changes WHEN those launches run, not what they compute: eight training steps
    (deterministic backward: bitwise), an evaluation pass and a state_dict read in between give identical embeddings,
    parameters, Adam moments and memory to ``bpr_step(...); opt.step()``."""
    from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
    cfg = SyntheticConfig("ft", 300, 25, 7000, 64, 2, 8, 2)
    g = make_graph(cfg, with_prices=False)
    d = g.data
    B, q, K = 64, 3, 8
    rs = np.random.RandomState(12)
    starts = [3000 + B * i for i in range(8)]
    negs = [rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * q) for _ in starts]

    def run(fused):
        torch.manual_seed(31)
        tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, DEV, n_layers=2, n_heads=2, dropout=0.1,
                    use_memory=use_memory, memory_dimension=64, message_function="identity", n_neighbors=K)
        tgn.deterministic = True
        opt = P.FusedAdam(tgn, lr=1e-3)
        dev = lambda a, t: tgn._to_dev(a, t)
        out = []
        for i, (s, neg) in enumerate(zip(starts, negs)):
            tgn.train()
            emb, b = tgn.embed_device(dev(d.sources[s:s + B], np.int32), dev(d.destinations[s:s + B], np.int32), [dev(neg, np.int32)], [q],
                                      dev(d.timestamps[s:s + B], np.float64), dev(d.edge_idxs[s:s + B], np.int32), K)
            if fused:
                loss = P.bpr_step(tgn, emb, b, q, optimizer=opt)
            else:
                loss = P.bpr_step(tgn, emb, b, q)
                opt.step()
            opt.zero_grad(set_to_none=True)
            if i == 3:                                   # readers of the parameters between two steps: state_dict joins by itself
                sd = {k: v.clone() for k, v in tgn.state_dict().items()}
                out.append(torch.cat([v.reshape(-1).float() for k, v in sorted(sd.items()) if v.is_floating_point()]))
            if i == 5:                                   # an evaluation pass (forward only) right behind a fused step
                tgn.eval()
                with torch.no_grad():
                    ev = torch.cat(tgn.compute_temporal_embeddings(d.sources[s:s + B], d.destinations[s:s + B], neg, d.timestamps[s:s + B],
                                                                   d.edge_idxs[s:s + B], K))
                out.append(ev.clone())
            out.append(emb.detach().clone())
            out.append(loss.detach().clone().reshape(1))
        tgn.join()
        torch.cuda.synchronize()
        out.append(tgn.flat_parameters.detach().clone())
        out.append(opt._m.clone()); out.append(opt._v.clone())
        if use_memory:
            out.append(tgn.memory.memory.detach().clone()); out.append(tgn.memory.msg_table.clone())
        return out

    a, b_ = run(False), run(True)
    assert len(a) == len(b_)
    for x, y in zip(a, b_):
        assert torch.equal(x, y)


# ---------------------------------------------------------------------------------------------
# pfo_segment_sum (the per-node sum of the instances' gradient rows: embedding_module.py:93-98 under autograd)
def _segment_sum_case(lens, W0, W1, by_pos, live_frac, with_seg_of, seed):
    import numpy as np
    from pfotgnrec_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(seed)
    lens = np.asarray(lens, np.int64)
    S, M = len(lens), int(lens.sum())
    N = M + 7                                                   # instances: more than the members (padding instances are in no segment)
    mem = rng.permutation(N)[:M].astype(np.int32)
    seg_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    live = (rng.rand(M) < live_frac).astype(np.uint8) if by_pos else None
    src0 = rng.randn(M if by_pos else N, W0).astype(np.float32)
    src1 = rng.randn(N, W1).astype(np.float32)
    rows0 = (src0 * live[:, None]) if by_pos else src0[mem]
    rows = np.concatenate([rows0, src1[mem]], 1).astype(np.float64)
    ref = np.zeros((S, W0 + W1))
    for s in range(S):
        ref[s] = rows[seg_ptr[s]:seg_ptr[s + 1]].sum(0)
    so = np.full((M // 16 + 2) * 16, -12345, np.int32)          # (the tail is never used: poison it)
    so[:M] = np.repeat(np.arange(S), lens)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_src0, d_src1, d_ptr, d_mem, d_so = t(src0), t(src1), t(seg_ptr), t(mem), t(so)
    d_live = t(live) if live is not None else None
    cap = S + 3
    out = torch.full((cap, W0 + W1), float("nan"), device=dev)
    n_rows = torch.tensor([S], dtype=torch.int32, device=dev)
    _lib.call("pfo_segment_sum", _lib.ptr(d_src0), W0, _lib.ptr(d_src1), W1, _lib.ptr(d_ptr), _lib.ptr(d_mem),
              _lib.ptr(d_so) if with_seg_of else None, M, _lib.ptr(n_rows), cap, 1 if by_pos else 0,
              _lib.ptr(d_live) if d_live is not None else None, _lib.ptr(out), _lib.stream_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isnan(got[S:]).all()                              # rows past *n_rows are not written
    mag = np.abs(rows).max() * max(1, lens.max())
    assert np.abs(got[:S] - ref).max() <= 4e-7 * mag, (np.abs(got[:S] - ref).max(), mag)
    return got[:S]


@pytest.mark.gpu
@pytest.mark.parametrize("with_seg_of", [False, True])
def test_segment_sum_ragged_empty_and_long_segments(with_seg_of):
    import numpy as np
    rng = np.random.RandomState(3)
    # empty segments (first, middle, last), singletons, one segment longer than a member batch (64) and than a chunk (16)
    lens = [0, 1, 3, 0, 0, 70, 1, 17, 16, 15, 2, 0] + list(rng.randint(0, 6, size=200)) + [130, 0]
    a = _segment_sum_case(lens, 704, 172, True, 0.3, with_seg_of, 1)      # the step's shape: flagged rows stored by position
    b = _segment_sum_case(lens, 704, 172, True, 0.3, with_seg_of, 1)
    assert np.array_equal(a, b)                                           # the same sum on every run
    _segment_sum_case(lens, 8, 4, False, 1.0, with_seg_of, 2)            # rows picked by member id, narrow rows
    _segment_sum_case(lens, 1200, 100, True, 1.0, with_seg_of, 3)        # wider than one 256-column pass
    _segment_sum_case([5], 704, 172, True, 1.0, with_seg_of, 4)
    _segment_sum_case(lens, 7, 5, False, 1.0, with_seg_of, 5)            # unaligned widths: the scalar kernel


@pytest.mark.gpu
def test_segment_sum_member_cut_equals_segment_cut_bitwise():
    import numpy as np
    lens = list(np.random.RandomState(9).randint(0, 45, size=300))
    a = _segment_sum_case(lens, 704, 172, True, 0.5, False, 7)
    b = _segment_sum_case(lens, 704, 172, True, 0.5, True, 7)
    assert np.array_equal(a, b)                                           # both add the rows of a segment in member order


# ------------------------------------------------------------------ the side streams keep hardware queues of their own
@pytest.mark.gpu
def test_side_stream_sits_in_a_priority_class_of_its_own():
    """with more streams in the process than GPU_MAX_HW_QUEUES (a process group brings RCCL's and c10d's), a
    normal-priority side stream shares a hardware queue with the caller's stream and every overlap of the step is lost.  The
    library's first side stream is created in the HIGH priority class (its own pool of hardware queues), and importing the
    package raises the number of hardware queues unless the caller set it."""
    import ctypes
    import pfotgnrec_amd  # noqa: F401
    from pfotgnrec_amd import _lib
    assert int(os.environ["GPU_MAX_HW_QUEUES"]) >= 8
    ptr = _lib.load().pfo_tgn_side_stream()
    assert ptr
    hip = ctypes.CDLL("libamdhip64.so")
    lo, hi = ctypes.c_int(0), ctypes.c_int(0)
    assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
    prio = ctypes.c_int(99)
    assert hip.hipStreamGetPriority(ctypes.c_void_p(ptr), ctypes.byref(prio)) == 0
    assert hi.value < lo.value, (lo.value, hi.value)           # the device has priority classes
    assert prio.value == hi.value, (prio.value, lo.value, hi.value)
    assert _lib.load().pfo_tgn_side_stream() == ptr             # one stream for the life of the process
