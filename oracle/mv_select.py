"""Oracle: mean-variance-efficient rank fusion (reference main.py:190-304).  Test infrastructure only.

Index-space restatement of the inline block: items are 0-based item indices
instead of stock-code strings and prices come from a dense ``[day, item, 30]``
array instead of the ``time_feature[day][code]`` dict, but every arithmetic
step uses the same numpy/scipy call as the reference (``np.log``, ``np.mean``,
``np.cov`` ddof=1, ``scipy.stats.rankdata`` average ranks, ``np.argsort``).
"""
import numpy as np
import scipy.stats as stats


def mv_scores_one(prices_day, cand_items, port_items, gamma):
    """y_mv for each candidate of one interaction (main.py:214-275).

    prices_day f64[I,30]; cand_items int[n_c]; port_items int[|P|] (empty = the ``'' in stocks_p`` branch).
    """
    cand_feature = prices_day[cand_items]
    cand_feature = np.log(cand_feature[:, 1:] / cand_feature[:, :-1])              # :218 / :227
    empty = len(port_items) == 0
    if not empty:
        port_feature = prices_day[port_items]
        port_feature = np.log(port_feature[:, 1:] / port_feature[:, :-1])          # :226
    y = []
    for feature in cand_feature:
        mu_i = np.mean(feature)                                                    # :243
        if empty:
            sigma_i = np.cov(feature)                                              # :247
            y_mv = (mu_i / gamma) / sigma_i                                        # :254
        else:
            cov_i = np.cov(feature, port_feature)                                  # :259
            sigma_ij = cov_i[0, 1:]
            sigma_i = cov_i[0, 0]
            n_holding = len(port_items)
            sum_sigma_ij = 1 / n_holding * np.sum(sigma_ij)                        # :268
            y_mv = (mu_i / gamma - 0.5 * sum_sigma_ij) / sigma_i                   # :271
        y.append(float(y_mv))
    return np.array(y, np.float64)


def fuse_ranks(y_mv, lambda_mv):
    """main.py:282-286: invest_rank (average ties), tgn_rank = [n..1], lambda blend (python floats)."""
    invest_rank = stats.rankdata(y_mv)
    n = len(y_mv)
    tgn_rank = stats.rankdata(np.arange(n)[::-1])
    new_rank = np.array([r1 * lambda_mv + r2 * (1 - lambda_mv) for r1, r2 in zip(invest_rank, tgn_rank)])
    return invest_rank, new_rank


def canonical_order(new_rank):
    """Tie policy of the build: stable ascending argsort, then reversed."""
    return np.argsort(new_rank, kind="stable")[::-1]


def mv_select(prices, day_idx, candidates, port_idx, port_len, gamma, lambda_mv, p_pos_num, p_neg_num,
              platform_order=False):
    """Whole block for a batch.  candidates i64[B, 1+C] 0-based item indices (column 0 = true destination).

    Returns p_pos i64[B,p], p_neg i64[B,q] (item indices), y_mv f64[B,1+C], new_rank f64[B,1+C].
    ``platform_order=True`` uses ``np.argsort(new_rank)[::-1]`` exactly as main.py:289 (ISA-dependent ties).
    """
    B, n_c = candidates.shape
    p_pos = np.zeros((B, p_pos_num), np.int64)
    p_neg = np.zeros((B, p_neg_num), np.int64)
    Y = np.zeros((B, n_c)); NR = np.zeros((B, n_c))
    for b in range(B):
        port = port_idx[b, :port_len[b]]
        y = mv_scores_one(prices[day_idx[b]], candidates[b], port, gamma)
        _, new_rank = fuse_ranks(y, lambda_mv)
        order = np.argsort(new_rank)[::-1] if platform_order else canonical_order(new_rank)
        sorted_items = candidates[b][order]
        p_pos[b] = sorted_items[:p_pos_num]                                        # :291
        p_neg[b] = sorted_items[-p_neg_num:]                                       # :292
        Y[b], NR[b] = y, new_rank
    return p_pos, p_neg, Y, NR
