"""Mean-variance-efficient positive/negative selection - the inline block main.py:190-304 as a
callable over the HIP kernel ``pfo_mv_select``.

Setup (host, once): prices ``[day, item, 30]`` -> log-returns ``[day, item, 29]`` with the same
``np.log(p[1:] / p[:-1])`` the reference applies per interaction (main.py:218,226-227), uploaded as
fp64.  Per batch everything runs on the GPU: y_mv per candidate, average-tie ranks, lambda blend,
canonical ordering (stable argsort reversed), top-p / bottom-q selection.
"""
import numpy as np

from . import _lib


def log_returns(prices):
    prices = np.asarray(prices, np.float64)
    return np.ascontiguousarray(np.log(prices[:, :, 1:] / prices[:, :, :-1]))


def prices_from_time_feature(time_feature, map_item_id, upper_u=None):
    """The reference's ``time_feature[day_key][stock] -> 30 prices`` dict (``time_feature_future_{p}.pkl``, main.py:88) ->
    (sorted day keys, f64[day, item, 30]).

    main.py looks a stock up by its CODE for portfolio members (``time_feature[t][p]``, main.py:223) and by its item NODE ID
    for candidates (``time_feature[t][c]``, main.py:217/224), so both kinds of keys are accepted: str keys go through
    ``map_item_id`` (code -> 0-based item index), integer keys are node ids (item index = id - upper_u - 1).  Items a day does
    not list keep a constant price of 1 (zero log-returns)."""
    days = sorted(time_feature.keys())
    n_items = len(map_item_id)
    first = next(iter(time_feature[days[0]].values()))
    arr = np.ones((len(days), n_items, len(first)), np.float64)
    for d, key in enumerate(days):
        for stock, p in time_feature[key].items():
            if isinstance(stock, str):
                if stock in map_item_id:
                    arr[d, map_item_id[stock]] = p
            elif upper_u is not None:
                i = int(stock) - int(upper_u) - 1
                if 0 <= i < n_items:
                    arr[d, i] = p
    return days, arr


def day_indices(timestamps, days):
    """main.py:212: ``t = str(ts)[:8]`` picks the trading day of an interaction; -> index into ``days`` (KeyError like the
    reference's dict lookup when the day is missing)."""
    pos = {str(k): i for i, k in enumerate(days)}
    return np.array([pos[str(ts)[:8]] for ts in np.asarray(timestamps).tolist()], np.int32)


class MVSampler:
    def __init__(self, prices, upper_u, device, gamma=2.0, lambda_mv=0.5, p_pos_num=1, p_neg_num=3, day_of=None):
        import torch
        _lib.require_gpu(device)
        self.device = torch.device(device)
        ret = log_returns(prices)
        self.n_days, self.n_items, self.n_ret = ret.shape
        self.returns = torch.from_numpy(ret).to(self.device)
        self.upper_u = int(upper_u)
        self.gamma, self.lambda_mv = float(gamma), float(lambda_mv)
        self.p_pos_num, self.p_neg_num = int(p_pos_num), int(p_neg_num)
        self.day_of = day_of

    def select_device(self, day_idx, cand, port_idx, port_len, want_scores=False):
        """day_idx i32[B]; cand i32[B,1+C] item node ids (col 0 = destination); port_idx i32[B,W]; port_len i32[B]."""
        import torch
        B, n_c = cand.shape
        W = port_idx.shape[1] if port_idx.dim() == 2 else 0
        p_pos = torch.empty((B, self.p_pos_num), dtype=torch.int32, device=self.device)
        p_neg = torch.empty((B, self.p_neg_num), dtype=torch.int32, device=self.device)
        y = torch.empty((B, n_c), dtype=torch.float64, device=self.device) if want_scores else None
        nr = torch.empty((B, n_c), dtype=torch.float64, device=self.device) if want_scores else None
        _lib.call("pfo_mv_select", _lib.ptr(self.returns), self.n_days, self.n_items, self.n_ret, _lib.ptr(day_idx),
                  _lib.ptr(cand), n_c, _lib.ptr(port_idx), _lib.ptr(port_len), W, B, self.upper_u, self.gamma,
                  self.lambda_mv, self.p_pos_num, self.p_neg_num, _lib.ptr(p_pos), _lib.ptr(p_neg), _lib.ptr(y),
                  _lib.ptr(nr), _lib.stream_ptr())
        return (p_pos, p_neg, y, nr) if want_scores else (p_pos, p_neg)

    def select(self, destinations_batch, negatives_batch, timestamps_batch, port_idx, port_len, want_scores=False):
        """Host-array entry mirroring main.py:207-304: returns flat i64 ``p_pos_batch`` / ``p_neg_batch`` node ids."""
        import torch
        from .rand_edge_sampler import _device_rows
        dst = np.asarray(destinations_batch, np.int64).reshape(-1, 1)
        cand = np.concatenate([dst, np.asarray(negatives_batch, np.int64)], 1)                    # main.py:207
        B, n_c = cand.shape
        day = np.asarray(self.day_of(timestamps_batch), np.int64)
        # one upload for (day, candidates), none for the portfolio rows when they are a batch of the packed dataset
        # (rand_edge_sampler.packed_portfolios_of / RandEdgeSampler.port_idx), one read-back for (p_pos, p_neg)
        both = torch.from_numpy(np.concatenate([day.reshape(-1), cand.reshape(-1)]).astype(np.int32)).to(self.device)
        pi, pl = _device_rows(np.asarray(port_idx) if not isinstance(port_idx, np.ndarray) else port_idx,
                              np.asarray(port_len) if not isinstance(port_len, np.ndarray) else port_len, self.device)
        if pi.dtype != torch.int32 or pl.dtype != torch.int32:
            pi, pl = pi.to(torch.int32), pl.to(torch.int32)
        out = self.select_device(both[:B], both[B:].view(B, n_c), pi, pl, want_scores)
        sel = torch.cat([out[0], out[1]], 1).cpu().numpy().astype(np.int64)
        res = [np.ascontiguousarray(sel[:, :self.p_pos_num]).reshape(-1), np.ascontiguousarray(sel[:, self.p_pos_num:]).reshape(-1)]
        res += [o.cpu().numpy() for o in out[2:]]
        return tuple(res)
