"""Deterministic synthetic user-stock interaction graphs.

The reference ships no data (README.md:45), so every configuration in
BASELINE.json is generated here from ``numpy.random.RandomState(seed)``:

* users ``1..U``, items ``U+1..U+I`` (node 0 / edge 0 are padding, App. A-1);
* timestamps: sorted ``randint(0, 2**24)`` cast to f64 - exact in f32, with
  duplicates so that the strict ``<`` of the sampler is exercised;
* ``edge_idx = 1..E``; edge features ``N(0,1)`` with row 0 zero; node features
  ``rand(n, D)`` (main.py:87);
* per-interaction portfolios: ``|P| ~ U{0..7}`` distinct stock codes, ``['']``
  when empty (main.py:214);
* prices ``100 * exp(cumsum(N(0, 0.02)))`` for ``n_days x I x 30``.

Nothing here touches the GPU; it is the shared input source for tests,
``bench.py`` and the oracle.
"""
from dataclasses import dataclass, field
import numpy as np

from .data import Data

TS_RANGE = 1 << 24


@dataclass
class SyntheticConfig:
    name: str
    n_users: int
    n_items: int
    n_edges: int
    dim: int
    n_layers: int
    n_neighbors: int
    n_heads: int
    edge_dim: int = 4
    use_memory: bool = True
    uniform: bool = False
    batch: int = 512
    n_days: int = 64
    seed: int = 1


CONFIGS = {
    # BASELINE.json configs[0..4]
    "C1": SyntheticConfig("C1", 1_000, 100, 10_000, 32, 1, 10, 2, batch=128),
    "C2": SyntheticConfig("C2", 50_000, 500, 1_000_000, 172, 2, 20, 2),
    "C3": SyntheticConfig("C3", 50_000, 500, 1_000_000, 172, 2, 20, 2),
    "C4": SyntheticConfig("C4", 500_000, 500, 10_000_000, 172, 2, 20, 2, batch=4096),
    "C5": SyntheticConfig("C5", 50_000, 500, 1_000_000, 172, 2, 20, 4, use_memory=False, uniform=True),
}


@dataclass
class SyntheticGraph:
    cfg: SyntheticConfig
    data: Data
    node_features: np.ndarray      # f64 [n, D]
    edge_features: np.ndarray      # f64 [E+1, Ef]
    upper_u: int
    map_item_id: dict              # stock code -> item index 0..I-1
    codes: list                    # item index -> stock code
    prices: np.ndarray             # f64 [n_days, I, 30]
    portfolio_idx: np.ndarray      # i32 [E, 8], -1 padded item indices (device-side form)
    portfolio_len: np.ndarray      # i32 [E]
    extra: dict = field(default_factory=dict)

    @property
    def n_nodes(self):
        return self.cfg.n_users + self.cfg.n_items + 1

    def day_of(self, ts):
        """Trading-day index of a timestamp (synthetic stand-in for ``str(ts)[:8]``, main.py:212)."""
        return (np.asarray(ts).astype(np.int64) * self.cfg.n_days) >> 24


def stock_code(i):
    return "%06d" % (i + 1)


def make_graph(cfg, with_prices=True, with_portfolios=True):
    rs = np.random.RandomState(cfg.seed)
    U, I, E = cfg.n_users, cfg.n_items, cfg.n_edges
    src = rs.randint(1, U + 1, size=E).astype(np.int64)
    dst = rs.randint(U + 1, U + I + 1, size=E).astype(np.int64)
    ts = np.sort(rs.randint(0, TS_RANGE, size=E)).astype(np.float64)
    eidx = np.arange(1, E + 1, dtype=np.int64)
    edge_feat = rs.randn(E + 1, cfg.edge_dim)
    edge_feat[0] = 0.0
    n = U + I + 1
    node_feat = rs.rand(n, cfg.dim)

    codes = [stock_code(i) for i in range(I)]
    map_item_id = {c: i for i, c in enumerate(codes)}

    pmax = 8
    if with_portfolios:
        plen = rs.randint(0, pmax, size=E).astype(np.int32)          # 0..7
        # distinct items per row: strictly increasing offsets (< 64 <= I) from a random base
        base = rs.randint(0, I, size=E)
        off = np.cumsum(rs.randint(1, 9, size=(E, pmax)), axis=1)
        pidx = ((base[:, None] + off) % I).astype(np.int32)
        pidx[np.arange(pmax)[None, :] >= plen[:, None]] = -1
    else:
        plen = np.zeros(E, np.int32)
        pidx = np.full((E, pmax), -1, np.int32)
    portfolios = np.empty(E, dtype=object)
    if E <= 200_000:
        for e in range(E):
            L = plen[e]
            portfolios[e] = [codes[j] for j in pidx[e, :L]] if L > 0 else [""]
    else:
        portfolios = None   # large graphs use the packed form only

    if with_prices:
        steps = rs.randn(cfg.n_days, I, 30) * 0.02
        prices = 100.0 * np.exp(np.cumsum(steps, axis=2))
    else:
        prices = np.zeros((0, I, 30))

    data = Data(src, dst, ts, eidx, labels=np.zeros(E, np.int64), portfolios=portfolios)
    return SyntheticGraph(cfg, data, node_feat, edge_feat, U, map_item_id, codes, prices, pidx, plen)


def split_train(graph, frac=0.8):
    """Chronological train split the way utils/data.py:28,50 does it (timestamp quantile)."""
    d = graph.data
    val_time = np.quantile(d.timestamps, frac)
    m = d.timestamps <= val_time
    return Data(d.sources[m], d.destinations[m], d.timestamps[m], d.edge_idxs[m],
                labels=None, portfolios=None if d.portfolios is None else d.portfolios[m])
