#!/usr/bin/env python3
"""Forward-only evaluation batch at C2 scale (evaluation.py:88-145: every interaction scores all items)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfotgnrec_amd as P
from pfotgnrec_amd.synthetic import CONFIGS, make_graph
cfg = CONFIGS["C2"]; g = make_graph(cfg, with_prices=False); d = g.data
dev = torch.device("cuda:0")
tgn = P.TGN(P.get_neighbor_finder(d, False), g.node_features, g.edge_features, dev, n_layers=2, n_heads=2, dropout=0.1,
            use_memory=True, memory_dimension=172, message_function="identity")
with torch.no_grad():
    tgn.memory.msg_table.normal_(0, 0.1); tgn.memory.memory.normal_(0, 0.1); tgn.memory.has_msg.fill_(1)
tgn.eval()
if len(sys.argv) > 1:
    tgn.eval_chunk_roots = int(sys.argv[1])          # roots per forward-only pass (default 16384)
B, n_items = 512, cfg.n_items
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
items = torch.arange(cfg.n_users + 1, cfg.n_users + 1 + n_items, dtype=torch.int32, device=dev).repeat(B)
times = []
for it in range(4):
    s = 900000 + it * B
    src, dst = t(d.sources[s:s + B], np.int32), t(d.destinations[s:s + B], np.int32)
    ts, ei = t(d.timestamps[s:s + B], np.float64), t(d.edge_idxs[s:s + B], np.int32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        emb, b = tgn.embed_device(src, dst, [items], [n_items], ts, ei, 20)
        rank, hits, ndcg = P.rank_metrics(emb, B, n_items)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
import json
print(json.dumps({"what": "evaluation batch (evaluation.py:63-145): forward only, every interaction scores all items, ranking on device",
                  "config": "C2 graph, L2 K20 D172 H2 memory+GRU", "interactions": B, "roots": B * (2 + n_items), "chunk_roots": tgn.eval_chunk_roots,
                  "ms_per_batch": round(1e3 * min(times), 2), "ms_all": [round(1e3 * x, 2) for x in times],
                  "interactions_per_s": round(B / min(times), 1), "root_embeddings_per_s": round(B * (2 + n_items) / min(times), 0),
                  "recall_at_5": round(hits[:, 2].mean().item(), 4)}))
