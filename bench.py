#!/usr/bin/env python3
"""Throughput of the TGN training step on MI355X (BASELINE.json metric: interactions/s, TGN fwd + BPR step).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched under torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic interactions, the loop body of
main.py:160-394: candidate-negative draw -> temporal neighbour sampling -> lazy memory update (GRU)
-> L-layer temporal graph attention -> BPR loss -> backward -> Adam -> memory persist + raw-message
store.  Workload at N=1: BASELINE.json configs[1] (C2: 50k users x 500 items, 1M edges, 2-layer
attention, 20 neighbours, dim 172, batch 512, 3 negatives).  Inputs are resident in HBM before the
timed region.  N > 1: edge-batch data parallelism, weak scaling (512 interactions per GPU per step,
global batch 512*N), one RCCL all-reduce of the flat gradient buffer per step.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     live HIP-event timing of the dominant kernel family over the timed region
  "cpu_baseline": the oracle (numpy restatement of the reference) timed on this host, bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32)
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the sparsity figure is never used)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--batch", type=int, default=0, help="interactions per GPU per step (default: the config's)")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=512, help="interactions per oracle step in the CPU baseline sample")
    ap.add_argument("--cpu-steps", type=int, default=1, help="timed oracle steps (one more runs first, untimed)")
    ap.add_argument("--no-prof", action="store_true", help="skip the HIP-event bracketing of kernel launches")
    ap.add_argument("--prof-every", type=int, default=8,
                    help="bracket the kernel launches of every Nth timed step with HIP events (the brackets cost ~0.2 ms "
                         "per step at C2, so the roofline sample is taken on a subset of the timed steps)")
    return ap.parse_args()


PMC_KERNEL = {"gemm_bx": "gemm_bx_areg_kernel", "gemm_tn_bx": "gemm_tn_group_bx_kernel",
              "gemm_bx_skinny": "gemm_bx_skinny_kernel",
              "gemm_nt": "void gemm_f32_kernel<false, false, 0, true>",
              "gemm_nn": "void gemm_f32_kernel<false, true, 0, true>",
              "gemm_tn": "void gemm_tn_group_kernel<true>",
              "attn_fwd": "void attn_fwd_kernel<3, 2>", "attn_bwd": "void attn_bwd_kernel<3, 2>"}


def pmc_traffic(family):
    """HBM bytes per launch of the family's kernel from the newest committed PMC pass (profiles/r*_summary.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs, (2*FETCH + WRITE)*1024 on gfx950); None if not profiled."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_summary.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["pmc"]["kernels"].get(PMC_KERNEL.get(family, ""))
        return int(k["hbm_bytes_per_launch_corrected"]) if k else None
    except Exception:
        return None


def steady_state_init(tgn, rs):
    """Every node holds a pending message and a non-zero memory, the state the reference reaches a few
    hundred batches into an epoch (#pending == #seen nodes -> n)."""
    import torch
    mem = tgn.memory
    with torch.no_grad():
        g = torch.Generator(device="cpu").manual_seed(1234)
        mem.memory.copy_((torch.randn(mem.memory.shape, generator=g) * 0.1).to(mem.memory.device))
        mem.msg_table.copy_((torch.randn(mem.msg_table.shape, generator=g) * 0.1).to(mem.memory.device))
        mem.msg_time.zero_()
        mem.last_update.zero_()
        mem.has_msg.fill_(1)
        mem.has_msg[0] = 0


def cpu_baseline(cfg, graph, batch, steps):
    """The oracle's full training step (same op list as the GPU step) on this host; numpy + BLAS threads."""
    from oracle import tgn_oracle as T
    from oracle.neighbor_finder import OracleNeighborFinder, build_adjacency
    d = graph.data
    onf = OracleNeighborFinder(*build_adjacency(d.sources, d.destinations, d.edge_idxs, d.timestamps), uniform=cfg.uniform)
    P = T.init_params(cfg.dim, cfg.edge_dim, cfg.n_layers, seed=0, use_memory=cfg.use_memory)
    ref = T.OracleTGN(onf, graph.node_features, graph.edge_features, P, cfg.n_layers, cfg.n_heads, cfg.use_memory)
    rs = np.random.RandomState(7)
    if cfg.use_memory:
        M = 3 * cfg.dim + cfg.edge_dim
        msgs = (rs.randn(graph.n_nodes, M) * 0.1).astype(np.float32)
        for v in range(1, graph.n_nodes):
            ref.messages[v] = [(msgs[v], np.float32(0))]
        ref.memory = (rs.randn(graph.n_nodes, cfg.dim) * 0.1).astype(np.float32)
    m = {k: np.zeros_like(v) for k, v in ref.P.items()}
    v2 = {k: np.zeros_like(v) for k, v in ref.P.items()}
    s = cfg.n_edges // 2
    times = []
    for it in range(steps + 1):
        sl = slice(s, s + batch)
        sb, db, tb, eb = d.sources[sl], d.destinations[sl], d.timestamps[sl], d.edge_idxs[sl]
        t0 = time.perf_counter()
        neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=batch * 3)
        se, de, ne = ref.compute_temporal_embeddings(sb, db, neg, tb, eb, cfg.n_neighbors)
        loss, cache = T.bpr_loss(se, de.reshape(batch, 1, -1), ne.reshape(batch, 3, -1))
        ds, dp, dn = T.bpr_loss_backward(cache)
        grads = ref.backward(np.concatenate([ds, dp.reshape(batch, -1), dn.reshape(3 * batch, -1)]))
        for k in ref.P:                                   # Adam, main.py:123
            g = grads[k]
            m[k] = 0.9 * m[k] + 0.1 * g
            v2[k] = 0.999 * v2[k] + 0.001 * g * g
            ref.P[k] = ref.P[k] - 1e-4 * (m[k] / (1 - 0.9 ** (it + 1))) / (np.sqrt(v2[k] / (1 - 0.999 ** (it + 1))) + 1e-8)
        dt = time.perf_counter() - t0
        if it > 0:                                        # first step pays page-in / BLAS warm-up
            times.append(dt)
        s += batch
    return batch / float(np.median(times)), float(np.sum(times))


def main():
    args = parse()
    import torch
    import pfotgnrec_amd as P
    from pfotgnrec_amd import _lib
    from pfotgnrec_amd.distributed import init_from_env, allreduce_flat_grad, broadcast_parameters
    from pfotgnrec_amd.synthetic import CONFIGS, make_graph
    import torch.distributed as dist

    rank, world, local = init_from_env()
    if world != args.gpus and world > 1:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if os.environ.get("PFO_FORCE_DEVICE") is not None:      # test hook: several ranks on one GPU
        local = int(os.environ["PFO_FORCE_DEVICE"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    cfg = CONFIGS[args.config]
    per_gpu = args.batch or cfg.batch
    B = per_gpu * world                                   # weak scaling: global batch grows with the GPU count
    ours = cfg.name == "C3"                               # C3 = C2 + MV-efficient sampler (main.py `ours` path)
    graph = make_graph(cfg, with_prices=ours)
    d = graph.data
    nf = P.get_neighbor_finder(d, uniform=cfg.uniform)
    tgn = P.TGN(nf, graph.node_features, graph.edge_features, dev, n_layers=cfg.n_layers, n_heads=cfg.n_heads,
                dropout=args.dropout, use_memory=cfg.use_memory, memory_dimension=cfg.dim, message_function="identity",
                n_neighbors=cfg.n_neighbors)
    tgn.set_data_parallel(rank, world)
    broadcast_parameters(tgn.flat_parameters, world)
    if cfg.use_memory:
        steady_state_init(tgn, None)
    opt = P.FusedAdam(tgn, lr=args.lr)

    # batch inputs resident in HBM before the timed region
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    src_all, dst_all = t(d.sources, np.int32), t(d.destinations, np.int32)
    ts_all, eidx_all = t(d.timestamps, np.float64), t(d.edge_idxs, np.int32)
    port_idx_all, port_len_all = t(graph.portfolio_idx, np.int32), t(graph.portfolio_len, np.int32)
    from pfotgnrec_amd.rand_edge_sampler import item_availability, DeviceNegativeSampler
    sampler = DeviceNegativeSampler(item_availability(d.destinations, graph.upper_u, cfg.n_items), graph.upper_u, dev, seed=1)
    n_neg = 3
    mvs = None
    if ours:
        mvs = P.MVSampler(graph.prices, graph.upper_u, dev, gamma=2.0, lambda_mv=0.5, p_pos_num=1, p_neg_num=3)
        day_all = t(graph.day_of(d.timestamps), np.int32)
    start = cfg.n_edges // 2                              # neighbourhoods are populated
    n_steps_total = args.warmup + args.steps
    span = cfg.n_edges - start - B                          # batches wrap inside the second half of the edge list
    assert span > 0, "batch larger than the timed half of the graph"

    def step(i):
        lo = start + (i * B) % span
        sl = slice(lo, lo + B)
        if mvs is None:
            neg = sampler.sample(port_idx_all[sl], port_len_all[sl], n_neg, offset=i)       # utils.py:86-114
            emb, b = tgn.embed_device(src_all[sl], dst_all[sl], [neg.reshape(-1)], [n_neg], ts_all[sl], eidx_all[sl],
                                      cfg.n_neighbors)                                      # tgn.py:219-327
            loss = P.bpr_loss(emb, b, n_neg, pos_block=1, grad_scale=1.0 / world)           # main.py:364-381
        else:
            cand_neg = sampler.sample(port_idx_all[sl], port_len_all[sl], 20, offset=i)     # main.py:194-195
            cand = torch.cat([dst_all[sl].unsqueeze(1), cand_neg], 1).contiguous()          # main.py:207
            p_pos, p_neg = mvs.select_device(day_all[sl], cand, port_idx_all[sl], port_len_all[sl])   # main.py:209-304
            emb, b = tgn.embed_device(src_all[sl], dst_all[sl], [p_pos.reshape(-1), p_neg.reshape(-1)], [1, 3], ts_all[sl],
                                      eidx_all[sl], cfg.n_neighbors)                        # tgn.py:102-217
            loss = P.bpr_loss(emb, b, n_neg, pos_block=2, grad_scale=1.0 / world)           # main.py:321-337
        loss.backward()                                                                     # main.py:388
        allreduce_flat_grad(tgn.flat_grad, world)
        opt.step()                                                                          # main.py:389
        opt.zero_grad(set_to_none=True)
        return loss

    tgn.train()
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    n_prof_steps = 0
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps_total):
        sampled = (not args.no_prof) and ((i - args.warmup) % max(1, args.prof_every) == 0)
        if sampled:
            _lib.prof_enable(True)
            n_prof_steps += 1
        loss = step(i)
        if sampled:
            _lib.prof_enable(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = _lib.prof_collect() if not args.no_prof else None
    _lib.prof_enable(False)
    final_loss = float(loss.detach())
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
        return
    value = args.steps * B / elapsed
    out = {
        "metric": "interactions/sec (TGN fwd+BPR step)", "value": round(value, 1), "unit": "interactions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d users x %d items, %d edges, TGN L%d K%d D%d H%d %s, %d negatives, batch %d/GPU"
                               % (cfg.name, cfg.n_users, cfg.n_items, cfg.n_edges, cfg.n_layers, cfg.n_neighbors, cfg.dim,
                                  cfg.n_heads, "memory+GRU" if cfg.use_memory else "no memory, uniform sampling", n_neg, per_gpu),
                   "global_batch": B, "parallelism": "dp%d" % world, "dropout": args.dropout, "final_loss": round(final_loss, 5)},
    }
    if prof is not None:
        # dominant kernel family by device time over the sampled steps.  Contractions on the bf16x3 kernels are priced
        # against the dense bf16 MFMA peak divided by the six piece products one fp32 product costs; the fp32-MFMA
        # kernels against the fp32 MFMA peak; the attention / sampling kernels against HBM.
        fam = {k: v for k, v in prof.items() if v["count"] > 0 and v["work"] > 0}
        dom = max(fam, key=lambda k: fam[k]["ms"])
        v = fam[dom]
        per_launch_s = v["ms"] / v["count"] * 1e-3
        if dom.startswith("gemm"):
            peak = MFMA_BF16_PEAK_TF / 6.0 if dom in ("gemm_bx", "gemm_tn_bx", "gemm_bx_skinny") else MFMA_F32_PEAK_TF
            achieved = v["work"] / v["count"] / per_launch_s / 1e12
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": None}
            if peak != MFMA_F32_PEAK_TF:
                roof["arithmetic"] = ("fp32 contraction as a 3-way bf16 operand split: 6 v_mfma_f32_16x16x32_bf16 per fp32 "
                                      "product block; peak = dense bf16 MFMA peak / 6")
        else:
            achieved = v["work"] / v["count"] / per_launch_s / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}
        roof["traffic"] = pmc_traffic(dom)
        roof["kernel"] = dom
        roof["avg_launch_us"] = round(per_launch_s * 1e6, 2)
        roof["launches"] = int(v["count"])
        roof["sampled_steps"] = n_prof_steps
        roof["families_ms_per_step"] = {k: round(x["ms"] / max(1, n_prof_steps), 4) for k, x in prof.items() if x["count"] > 0}
        tot = lambda ks: sum(prof[k]["work"] for k in ks) / max(1e-9, sum(prof[k]["ms"] for k in ks) * 1e-3)
        roof["gemm_all_tflops"] = round(tot(["gemm_nt", "gemm_nn", "gemm_tn", "gemm_bx", "gemm_tn_bx", "gemm_bx_skinny"]) / 1e12, 2)
        roof["attn_all_gbs"] = round(tot(["attn_fwd", "attn_bwd"]) / 1e9, 1)
        out["roofline"] = roof
    if not args.no_cpu_baseline and world == 1:
        try:
            threads = os.cpu_count() or 1
            try:                                            # threads the BLAS behind numpy actually uses
                from threadpoolctl import threadpool_info
                threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
            except Exception:
                pass
            v, spent = cpu_baseline(cfg, graph, args.cpu_batch, args.cpu_steps)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "interactions/s", "cores": threads, "kind": "port",
                                   "sample": "%d step(s) of %d interactions of the same workload after one untimed step "
                                             "(oracle/tgn_oracle.py: numpy fp32 + BLAS + C fmaf/cosf helper, steady-state "
                                             "memory), %.1f s timed" % (args.cpu_steps, args.cpu_batch, spent)}
        except Exception as e:  # the baseline never blocks the measurement
            out["cpu_baseline"] = {"value": None, "unit": "interactions/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()
