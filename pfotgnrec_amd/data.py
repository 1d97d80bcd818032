"""Interaction-table container mirroring the reference's ``Data`` (utils/data.py:6-16).

Only the fields the training path reads are kept; ``labels`` is accepted for
signature compatibility.  Arrays are host numpy, exactly as ``main.py`` slices
them per batch (main.py:183-187).
"""
import numpy as np


class Data:
    def __init__(self, sources, destinations, timestamps, edge_idxs, labels=None, portfolios=None):
        self.sources = np.asarray(sources)
        self.destinations = np.asarray(destinations)
        self.timestamps = np.asarray(timestamps)
        self.edge_idxs = np.asarray(edge_idxs)
        self.labels = labels
        self.n_interactions = len(self.sources)
        self.unique_nodes = set(self.sources.tolist()) | set(self.destinations.tolist())
        self.n_unique_nodes = len(self.unique_nodes)
        self.portfolios = portfolios


def compute_time_statistics(sources, destinations, timestamps):
    """Mean/std of per-node inter-event gaps (utils/data.py:75-99), vectorised.

    The reference walks the edge list with two dicts; the same numbers come out
    of a stable sort by node followed by a diff with a zero first element.
    """
    def _gaps(ids, ts):
        order = np.argsort(ids, kind="stable")
        ids_s, ts_s = ids[order], ts[order].astype(np.float64)
        prev = np.concatenate([[0.0], ts_s[:-1]])
        first = np.concatenate([[True], ids_s[1:] != ids_s[:-1]])
        prev[first] = 0.0
        gaps = np.empty_like(ts_s)
        gaps[order] = ts_s - prev
        return gaps
    sources = np.asarray(sources)
    destinations = np.asarray(destinations)
    timestamps = np.asarray(timestamps)
    g_src = _gaps(sources, timestamps)
    g_dst = _gaps(destinations, timestamps)
    return float(np.mean(g_src)), float(np.std(g_src)), float(np.mean(g_dst)), float(np.std(g_dst))


def get_data(dataset_name, period, root="./data"):
    """Reads the reference's on-disk format and applies its chronological 80/10/10 split (utils/data.py:18-72).

    Files under ``{root}/period_{period}/`` (README.md:45): ``ml_{name}.json`` (records u, i, ts, label, idx, portfolio),
    ``ml_{name}.npy`` (edge features, row 0 = padding), ``ml_{name}_node.npy`` (node features).
    Returns (node_features, edge_features, full_data, train_data, val_data, test_data, upper_u) like the reference.
    """
    import os
    import pandas as pd
    base = os.path.join(root, "period_{}".format(period))
    graph_df = pd.read_json(os.path.join(base, "ml_{}.json".format(dataset_name)))
    edge_features = np.load(os.path.join(base, "ml_{}.npy".format(dataset_name)))
    node_features = np.load(os.path.join(base, "ml_{}_node.npy".format(dataset_name)))
    val_time, test_time = list(np.quantile(graph_df.ts, [0.8, 0.9]))                       # data.py:28
    sources, destinations = graph_df.u.values, graph_df.i.values
    edge_idxs, labels, timestamps = graph_df.idx.values, graph_df.label.values, graph_df.ts.values
    portfolios = graph_df.portfolio.values
    full_data = Data(sources, destinations, timestamps, edge_idxs, labels, portfolios)
    train_mask = timestamps <= val_time                                                    # data.py:50-52
    val_mask = np.logical_and(timestamps <= test_time, timestamps > val_time)
    test_mask = timestamps > test_time
    pick = lambda m: Data(sources[m], destinations[m], timestamps[m], edge_idxs[m], labels[m], portfolios[m])
    upper_u = graph_df.u.max()                                                             # data.py:59
    return node_features, edge_features, full_data, pick(train_mask), pick(val_mask), pick(test_mask), upper_u
