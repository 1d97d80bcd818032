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
