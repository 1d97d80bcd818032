"""Temporal neighbour finder - host mirror of the reference's ``NeighborFinder`` /
``get_neighbor_finder`` (utils/utils.py:117-219) over the HIP lookup kernel ``pfo_tnbr_sample``.

The adjacency is a time-sorted CSR (``indptr i64[n+1]``, ``nbr i32``, ``eidx i32``, ``ts f64``)
built once on the host and kept resident in HBM; every lookup runs on the GPU.  There is no CPU
lookup path.
"""
import numpy as np

from . import _lib


def _dev_key(device):
    """'cuda' and 'cuda:<current>' name the same device: one cache entry for both spellings."""
    import torch
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


def build_csr(sources, destinations, edge_idxs, timestamps, max_node_idx=None):
    """Per-node adjacency, stable-sorted by timestamp (utils/utils.py:117-142).

    Each edge contributes ``(dst, eidx, ts)`` to its source's row and ``(src, eidx, ts)`` to its
    destination's row, in edge order; rows are sorted by timestamp with ties keeping that order
    (the reference uses Python's stable ``sorted``).  One stable lexsort by (owner, ts) does it.
    """
    sources = np.asarray(sources, np.int64)
    destinations = np.asarray(destinations, np.int64)
    edge_idxs = np.asarray(edge_idxs, np.int64)
    timestamps = np.asarray(timestamps, np.float64)
    if not (len(sources) == len(destinations) == len(edge_idxs) == len(timestamps)):
        raise ValueError("sources, destinations, edge_idxs and timestamps must have equal length")
    if max_node_idx is None:
        max_node_idx = int(max(sources.max(), destinations.max())) if len(sources) else 0
    E = len(sources)
    owner = np.empty(2 * E, np.int64)
    other = np.empty(2 * E, np.int64)
    owner[0::2], owner[1::2] = sources, destinations
    other[0::2], other[1::2] = destinations, sources
    eid = np.repeat(edge_idxs, 2)
    ts = np.repeat(timestamps, 2)
    order = np.lexsort((ts, owner))
    counts = np.bincount(owner, minlength=max_node_idx + 1)
    indptr = np.zeros(max_node_idx + 2, np.int64)
    np.cumsum(counts, out=indptr[1:])
    if E and (other.max() >= 2 ** 31 or eid.max() >= 2 ** 31):
        raise ValueError("node / edge ids must fit in int32")
    return indptr, other[order].astype(np.int32), eid[order].astype(np.int32), ts[order]


def build_csr_device(sources, destinations, edge_idxs, timestamps, device, max_node_idx=None):
    """Same adjacency as ``build_csr`` built on the GPU by the native stable radix sort ``pfo_csr_build`` (SURVEY §8f-2;
    csrc/csr.hip): the reference's Python build takes 3.8 s per million edges, the host lexsort ~1 s, this milliseconds.
    Returns device tensors (indptr i64[n+1], nbr i32[2E], eidx i32[2E], ts f64[2E])."""
    import torch
    _lib.require_gpu(device)
    dev = torch.device(device)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(np.asarray(a)), dtype=dt, device=dev).contiguous()
    if len(sources) and (int(np.max(sources)) >= 2 ** 31 or int(np.max(destinations)) >= 2 ** 31 or int(np.max(edge_idxs)) >= 2 ** 31):
        raise ValueError("node / edge ids must fit in int32")
    src, dst, eid, ts = t(sources, torch.int32), t(destinations, torch.int32), t(edge_idxs, torch.int32), t(timestamps, torch.float64)
    E = int(src.shape[0])
    if max_node_idx is None:
        max_node_idx = int(max(int(np.max(sources)), int(np.max(destinations)))) if E else 0
    _check_ids(sources, destinations, max_node_idx)
    return _csr_build_native(src, dst, eid, ts, max_node_idx + 1, dev)


def _check_ids(sources, destinations, max_node_idx):
    """The device build sorts entries by the low bits of the owner id and counts only owners inside the table: an id outside
    [0, max_node_idx] would yield a silently corrupt CSR (the host build raises from np.bincount / the index store)."""
    if len(sources) == 0:
        return
    lo = min(int(np.min(sources)), int(np.min(destinations)))
    hi = max(int(np.max(sources)), int(np.max(destinations)))
    if lo < 0 or hi > max_node_idx:
        raise ValueError("node ids must lie in [0, %d]: found %d .. %d" % (max_node_idx, lo, hi))


def _csr_build_native(src, dst, eid, ts, n_nodes, dev):
    import torch
    E = int(src.shape[0])
    nbytes = _lib.load().pfo_csr_build_workspace_bytes(E, n_nodes)
    ws = torch.empty(max(1, nbytes), dtype=torch.uint8, device=dev)
    indptr = torch.empty(n_nodes + 1, dtype=torch.int64, device=dev)
    nbr = torch.empty(2 * E, dtype=torch.int32, device=dev)
    eidx = torch.empty(2 * E, dtype=torch.int32, device=dev)
    tss = torch.empty(2 * E, dtype=torch.float64, device=dev)
    _lib.call("pfo_csr_build", _lib.ptr(src), _lib.ptr(dst), _lib.ptr(eid), _lib.ptr(ts), E, n_nodes, _lib.ptr(indptr),
              _lib.ptr(nbr), _lib.ptr(eidx), _lib.ptr(tss), _lib.ptr(ws), nbytes, _lib.stream_ptr())
    return indptr, nbr, eidx, tss


class NeighborFinder:
    """Drop-in for utils/utils.py:130.  ``adj_list`` is the reference's list of per-node
    ``[(neighbor, edge_idx, timestamp), ...]`` lists; ``from_arrays`` skips that detour.

    The host arrays ``indptr / nbr / eidx / ts`` are MIRRORS of the device CSR: after a device-side build or ``append`` they
    are stale and are fetched only when something reads them (a 10 M-edge adjacency is 0.5 GB of D2H per refresh)."""

    def __init__(self, adj_list=None, uniform=False, seed=None, _csr=None, _dev_csr=None, _device=None):
        self._host = None
        self._dev = {}
        if _dev_csr is not None:
            self._dev[_dev_key(_device)] = _dev_csr
            self.n_nodes = int(_dev_csr[0].shape[0]) - 1
            self._max_nbr, self._max_eidx = None, None
        else:
            if _csr is None:
                owner, nbr, eidx, ts = [], [], [], []
                for node, lst in enumerate(adj_list):
                    lst = sorted(lst, key=lambda x: x[2])
                    owner.extend([node] * len(lst))
                    nbr.extend(x[0] for x in lst)
                    eidx.extend(x[1] for x in lst)
                    ts.extend(x[2] for x in lst)
                counts = np.bincount(np.asarray(owner, np.int64), minlength=len(adj_list))
                indptr = np.zeros(len(adj_list) + 1, np.int64)
                np.cumsum(counts, out=indptr[1:])
                _csr = (indptr, np.asarray(nbr, np.int32), np.asarray(eidx, np.int32), np.asarray(ts, np.float64))
            self._host = tuple(_csr)
            self.n_nodes = len(self._host[0]) - 1
            self._max_nbr = int(np.max(self._host[1], initial=0))
            self._max_eidx = int(np.max(self._host[2], initial=0))
        self.uniform = uniform
        self.seed = 0 if seed is None else int(seed)
        self._calls = 0
        self._version = 0          # bumped whenever the adjacency changes (append): consumers re-fetch the device arrays

    # ---- host mirrors (lazy)
    def _host_arrays(self):
        if self._host is None:
            dev_csr = next(iter(self._dev.values()))
            self._host = tuple(a.cpu().numpy() for a in dev_csr)
        return self._host

    indptr = property(lambda self: self._host_arrays()[0])
    nbr = property(lambda self: self._host_arrays()[1])
    eidx = property(lambda self: self._host_arrays()[2])
    ts = property(lambda self: self._host_arrays()[3])

    def max_neighbor_id(self):
        """Largest neighbour id in the adjacency (tracked incrementally; one device reduction after a device-side build)."""
        if self._max_nbr is None:
            a = next(iter(self._dev.values()))[1]
            self._max_nbr = int(a.max().item()) if a.numel() else 0
        return self._max_nbr

    def max_edge_idx(self):
        if self._max_eidx is None:
            a = next(iter(self._dev.values()))[2]
            self._max_eidx = int(a.max().item()) if a.numel() else 0
        return self._max_eidx

    def rows_end(self, n_rows):
        """indptr[n_rows] and indptr[-1] without materialising the host mirror."""
        if self._host is not None:
            return int(self._host[0][n_rows]), int(self._host[0][-1])
        p = next(iter(self._dev.values()))[0]
        return int(p[n_rows].item()), int(p[-1].item())

    @classmethod
    def from_arrays(cls, sources, destinations, edge_idxs, timestamps, uniform=False, max_node_idx=None, seed=None,
                    device=None):
        """``device`` given: build the CSR on that GPU and keep it there (host copies are made lazily on request)."""
        if device is None:
            return cls(uniform=uniform, seed=seed, _csr=build_csr(sources, destinations, edge_idxs, timestamps, max_node_idx))
        dev_csr = build_csr_device(sources, destinations, edge_idxs, timestamps, device, max_node_idx)
        obj = cls(uniform=uniform, seed=seed, _dev_csr=dev_csr, _device=device)
        if len(sources):                       # the extrema are known from the inputs: no device read-back later
            obj._max_nbr = int(max(int(np.max(sources)), int(np.max(destinations))))
            obj._max_eidx = int(np.max(edge_idxs))
        return obj

    def device_arrays(self, device):
        """CSR tensors resident on ``device`` (uploaded once)."""
        import torch
        key = _dev_key(device)
        if key not in self._dev:
            _lib.require_gpu(device)
            self._dev[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in self._host_arrays())
        return self._dev[key]

    def append(self, sources, destinations, edge_idxs, timestamps, device=None):
        """Adds new interactions to the adjacency in place (the reference can only rebuild, 3.8 s per million
        edges).  Equal to a rebuild over [old edges ; new edges]: rows stay sorted by timestamp, a new entry goes behind
        every older entry with the same timestamp.  The new edges are sorted on the device (``pfo_csr_build``) and merged
        row by row (``pfo_csr_append``); node ids beyond the current table grow it."""
        import torch
        _lib.require_gpu(device)
        if device is None:
            device = next(iter(self._dev)) if self._dev else "cuda"
        dev = torch.device(device)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(np.asarray(a)), dtype=dt, device=dev).contiguous()
        src, dst = t(sources, torch.int32), t(destinations, torch.int32)
        eid, ts = t(edge_idxs, torch.int32), t(timestamps, torch.float64)
        m = int(src.shape[0])
        if m == 0:
            return self
        n_nodes = max(self.n_nodes, int(max(int(np.max(sources)), int(np.max(destinations)))) + 1)
        _check_ids(sources, destinations, n_nodes - 1)
        if int(np.max(edge_idxs)) >= 2 ** 31:
            raise ValueError("edge ids must fit in int32")
        o_ptr, o_nbr, o_eid, o_ts = self.device_arrays(dev)
        a_ptr, a_nbr, a_eid, a_ts = _csr_build_native(src, dst, eid, ts, n_nodes, dev)
        total = int(o_nbr.shape[0]) + 2 * m
        n_ptr = torch.empty(n_nodes + 1, dtype=torch.int64, device=dev)
        n_nbr = torch.empty(total, dtype=torch.int32, device=dev)
        n_eid = torch.empty(total, dtype=torch.int32, device=dev)
        n_ts = torch.empty(total, dtype=torch.float64, device=dev)
        _lib.call("pfo_csr_append", _lib.ptr(o_ptr), _lib.ptr(o_nbr), _lib.ptr(o_eid), _lib.ptr(o_ts), self.n_nodes,
                  _lib.ptr(a_ptr), _lib.ptr(a_nbr), _lib.ptr(a_eid), _lib.ptr(a_ts), n_nodes, _lib.ptr(n_ptr), _lib.ptr(n_nbr),
                  _lib.ptr(n_eid), _lib.ptr(n_ts), _lib.stream_ptr())
        # extrema tracked from the appended batch; the host mirrors go stale and are refetched only if somebody reads them
        self._max_nbr = max(self.max_neighbor_id(), int(max(int(np.max(sources)), int(np.max(destinations)))))
        self._max_eidx = max(self.max_edge_idx(), int(np.max(edge_idxs)))
        self._dev = {_dev_key(dev): (n_ptr, n_nbr, n_eid, n_ts)}
        self._host = None
        self.n_nodes = n_nodes
        self._version += 1
        return self

    def next_stream_offset(self):
        self._calls += 1
        return self._calls << 20

    def get_temporal_neighbor(self, source_nodes, timestamps, n_neighbors=20, draws=None, device=None):
        """utils/utils.py:163-219 -> (neighbors i32[N,K], edge_idxs i32[N,K], edge_times f32[N,K]) numpy.

        Most-recent mode is index-exact with the reference.  Uniform mode draws with Philox unless
        ``draws`` (i64[N,K] positions into each query's history, the reference's ``sampled_idx``) is given.
        """
        import torch
        assert len(source_nodes) == len(timestamps)
        _lib.require_gpu(device)
        device = torch.device("cuda") if device is None else torch.device(device)
        N = len(source_nodes)
        K = n_neighbors if n_neighbors > 0 else 1
        if n_neighbors <= 0 or N == 0:                      # utils.py:175: one all-padding column
            return (np.zeros((N, K), np.int32), np.zeros((N, K), np.int32), np.zeros((N, K), np.float32))
        indptr, nbr, eidx, ts = self.device_arrays(device)
        q_nodes = torch.from_numpy(np.ascontiguousarray(source_nodes, dtype=np.int64).astype(np.int32)).to(device)
        q_ts = torch.from_numpy(np.ascontiguousarray(timestamps, dtype=np.float64)).to(device)
        o_nbr = torch.empty((N, K), dtype=torch.int32, device=device)
        o_eidx = torch.empty((N, K), dtype=torch.int32, device=device)
        o_et = torch.empty((N, K), dtype=torch.float32, device=device)
        mode, d_draws = 0, None
        if self.uniform:
            mode = 2
            if draws is not None:
                mode = 1
                d_draws = torch.from_numpy(np.ascontiguousarray(draws, dtype=np.int64)).to(device)
        _lib.call("pfo_tnbr_sample", _lib.ptr(indptr), _lib.ptr(nbr), _lib.ptr(eidx), _lib.ptr(ts), self.n_nodes,
                  _lib.ptr(q_nodes), _lib.ptr(q_ts), N, K, mode, _lib.ptr(d_draws), self.seed, self.next_stream_offset(),
                  _lib.ptr(o_nbr), _lib.ptr(o_eidx), _lib.ptr(o_et), None, None, None, _lib.stream_ptr())
        return o_nbr.cpu().numpy(), o_eidx.cpu().numpy(), o_et.cpu().numpy()


def get_neighbor_finder(data, uniform, max_node_idx=None):
    """utils/utils.py:117-127."""
    return NeighborFinder.from_arrays(data.sources, data.destinations, data.edge_idxs, data.timestamps,
                                      uniform=uniform, max_node_idx=max_node_idx)
