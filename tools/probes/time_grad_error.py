#!/usr/bin/env python3
"""Relative L2 error of the time-encoder gradients against the oracle over the steps of tests/test_gpu_tgn_step.py's
oracle loop, repeated (the level-0 scatter and the fp64 bins use atomics: the error varies in its last digits run to run).
Usage: python tools/probes/time_grad_error.py D H L K use_mem repeats"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pfotgnrec_amd as P
from pfotgnrec_amd.synthetic import SyntheticConfig, make_graph
from oracle import tgn_oracle as T
from oracle.neighbor_finder import OracleNeighborFinder, build_adjacency

D, H, L, K, use_mem, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == "1", int(sys.argv[6])
DEV = torch.device("cuda:0")
worst = {}
for rep in range(reps):
    torch.manual_seed(1234 + D + H)
    cfg = SyntheticConfig("t", 300, 25, 5000, D, L, K, H)
    g = make_graph(cfg, with_prices=False); d = g.data
    tgn = P.TGN(P.get_neighbor_finder(d, uniform=False), g.node_features, g.edge_features, DEV, n_layers=L, n_heads=H, dropout=0.0,
                use_memory=use_mem, memory_dimension=D, message_function="identity", n_neighbors=K)
    with torch.no_grad():
        tgn.time_encoder.w.bias.normal_(0, 0.3)
        for att in tgn.embedding_module.attention_models:
            att.multi_head_target.in_proj_bias.normal_(0, 0.1); att.multi_head_target.out_proj.bias.normal_(0, 0.1)
    opt = P.FusedAdam(tgn, lr=1e-3)
    onf = OracleNeighborFinder(*build_adjacency(d.sources, d.destinations, d.edge_idxs, d.timestamps), uniform=False)
    names = [k for k in tgn.state_dict() if "layer_norm" not in k and not k.startswith("memory.")]
    ref = T.OracleTGN(onf, g.node_features, g.edge_features, {k: tgn.state_dict()[k].cpu().numpy() for k in names}, L, H, use_mem)
    rs = np.random.RandomState(5); B = 40
    for step in range(4):
        s = 2500 + step * B
        sb, db, tb, eb = d.sources[s:s + B], d.destinations[s:s + B], d.timestamps[s:s + B], d.edge_idxs[s:s + B]
        neg = rs.randint(cfg.n_users + 1, cfg.n_users + cfg.n_items + 1, size=B * 3)
        ref.P = {k: tgn.state_dict()[k].detach().cpu().numpy().copy() for k in names}
        tgn.train(); opt.zero_grad()
        se, de, ne = tgn.compute_temporal_embeddings(sb, db, neg, tb, eb, K)
        rse, rde, rne = ref.compute_temporal_embeddings(sb, db, neg, tb, eb, K)
        loss = P.bpr_loss(torch.cat([se, de, ne]), B, 3); loss.backward()
        rl, cache = T.bpr_loss(rse, rde.reshape(B, 1, -1), rne.reshape(B, 3, -1))
        ds, dp, dn = T.bpr_loss_backward(cache)
        rgrads = ref.backward(np.concatenate([ds, dp.reshape(B, -1), dn.reshape(3 * B, -1)]))
        zmin = float(np.abs(ref._ctx[4]["z1"]).min())              # top layer's fc1 pre-activation closest to the ReLU kink (oracle)
        for name, p in tgn.named_parameters():
            if name in rgrads and p.grad is not None and np.abs(rgrads[name]).max() > 1e-7:
                r = rgrads[name].reshape(p.shape); got = p.grad.cpu().numpy().astype(np.float64)
                e = np.linalg.norm(got - r) / (np.linalg.norm(r) + 1e-30)
                worst[name] = max(worst.get(name, 0.0), e)
                if e > (1e-4 if name.startswith("time_encoder") else 3e-3):
                    print("OUTLIER rep %d step %d %-40s %.2e   top-layer min|z1| %.2e" % (rep, step, name, e, zmin))
        if rep == 0:
            print("step %d top-layer min|z1| %.2e" % (step, zmin))
        opt.step()
print("worst:", {k.split(".")[-2] + "." + k.split(".")[-1]: "%.1e" % v for k, v in worst.items()})
