"""Oracle: temporal neighbour sampler (reference utils/utils.py:117-219).  Test infrastructure only.

Same data structures and numpy calls as the reference so that, with the same
``np.random.seed``, the uniform mode consumes the global MT19937 stream in the
same order and reproduces the reference bit for bit.
"""
import numpy as np


def build_adjacency(sources, destinations, edge_idxs, timestamps, max_node_idx=None):
    """utils/utils.py:117-127 + 131-142: per-node lists, stable-sorted by timestamp.

    The reference appends ``(dst, eidx, ts)`` to ``adj[src]`` then ``(src, eidx,
    ts)`` to ``adj[dst]`` for each edge in order and sorts each list with
    Python's stable ``sorted(key=ts)``.  A stable lexsort of the interleaved
    entry list by (node, ts) yields the same per-node order.
    """
    sources = np.asarray(sources, np.int64)
    destinations = np.asarray(destinations, np.int64)
    edge_idxs = np.asarray(edge_idxs, np.int64)
    timestamps = np.asarray(timestamps, np.float64)
    if max_node_idx is None:
        max_node_idx = int(max(sources.max(), destinations.max()))
    E = len(sources)
    owner = np.empty(2 * E, np.int64); owner[0::2] = sources; owner[1::2] = destinations
    other = np.empty(2 * E, np.int64); other[0::2] = destinations; other[1::2] = sources
    eid = np.repeat(edge_idxs, 2)
    ts = np.repeat(timestamps, 2)
    order = np.lexsort((ts, owner))          # stable: ties keep append order
    owner, other, eid, ts = owner[order], other[order], eid[order], ts[order]
    counts = np.bincount(owner, minlength=max_node_idx + 1)
    indptr = np.zeros(max_node_idx + 2, np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr, other, eid, ts


class OracleNeighborFinder:
    def __init__(self, indptr, nbr, eidx, ts, uniform=False):
        self.indptr, self.nbr, self.eidx, self.ts = indptr, nbr, eidx, ts
        self.uniform = uniform

    @classmethod
    def from_data(cls, data, uniform, max_node_idx=None):
        return cls(*build_adjacency(data.sources, data.destinations, data.edge_idxs,
                                    data.timestamps, max_node_idx), uniform=uniform)

    def find_before(self, src_idx, cut_time):
        """utils/utils.py:150-161 - entries with ts strictly < cut_time (searchsorted side='left')."""
        lo, hi = self.indptr[src_idx], self.indptr[src_idx + 1]
        row_ts = self.ts[lo:hi]
        i = np.searchsorted(row_ts, cut_time)
        return self.nbr[lo:lo + i], self.eidx[lo:lo + i], row_ts[:i]

    def get_temporal_neighbor(self, source_nodes, timestamps, n_neighbors=20, draw_log=None):
        """utils/utils.py:163-219.  ``draw_log`` (list) records uniform-mode ``sampled_idx`` rows."""
        assert len(source_nodes) == len(timestamps)
        tmp = n_neighbors if n_neighbors > 0 else 1
        N = len(source_nodes)
        neighbors = np.zeros((N, tmp)).astype(np.int32)
        edge_times = np.zeros((N, tmp)).astype(np.float32)
        edge_idxs = np.zeros((N, tmp)).astype(np.int32)
        for i, (node, t) in enumerate(zip(source_nodes, timestamps)):
            s_nbr, s_eidx, s_ts = self.find_before(int(node), t)
            if len(s_nbr) > 0 and n_neighbors > 0:
                if self.uniform:
                    sampled_idx = np.random.randint(0, len(s_nbr), n_neighbors)      # :194 global RNG
                    if draw_log is not None:
                        draw_log.append((i, len(s_nbr), sampled_idx.copy()))
                    neighbors[i, :] = s_nbr[sampled_idx]
                    edge_times[i, :] = s_ts[sampled_idx]
                    edge_idxs[i, :] = s_eidx[sampled_idx]
                    pos = edge_times[i, :].argsort()                                 # :201 default (unstable) sort on f32
                    neighbors[i, :] = neighbors[i, :][pos]
                    edge_times[i, :] = edge_times[i, :][pos]
                    edge_idxs[i, :] = edge_idxs[i, :][pos]
                else:
                    s_ts, s_nbr, s_eidx = s_ts[-n_neighbors:], s_nbr[-n_neighbors:], s_eidx[-n_neighbors:]
                    neighbors[i, n_neighbors - len(s_nbr):] = s_nbr
                    edge_times[i, n_neighbors - len(s_ts):] = s_ts
                    edge_idxs[i, n_neighbors - len(s_eidx):] = s_eidx
        return neighbors, edge_idxs, edge_times

    def gather_uniform(self, source_nodes, timestamps, sampled_idx, n_neighbors):
        """Uniform mode with INJECTED draws and the canonical (stable) time re-sort.

        ``sampled_idx`` is i64[N, K]; rows of queries without history are ignored.
        """
        N = len(source_nodes)
        neighbors = np.zeros((N, n_neighbors), np.int32)
        edge_times = np.zeros((N, n_neighbors), np.float32)
        edge_idxs = np.zeros((N, n_neighbors), np.int32)
        for i, (node, t) in enumerate(zip(source_nodes, timestamps)):
            s_nbr, s_eidx, s_ts = self.find_before(int(node), t)
            if len(s_nbr) > 0:
                sel = sampled_idx[i]
                et = s_ts[sel].astype(np.float32)
                pos = np.argsort(et, kind="stable")
                neighbors[i] = s_nbr[sel][pos]
                edge_times[i] = et[pos]
                edge_idxs[i] = s_eidx[sel][pos]
        return neighbors, edge_idxs, edge_times
