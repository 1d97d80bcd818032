"""Oracle: candidate-negative draw (reference utils/utils.py:65-114).  Test infrastructure only."""
import numpy as np


class OracleRandEdgeSampler:
    def __init__(self, src_list, dst_list, portfolio_list, upper_u, map_item_id, seed=None):
        self.seed = None
        self.src_list = src_list
        self.dst_unique = np.unique(dst_list)                                      # :73
        pl = [[map_item_id[item] for item in sub if item] for sub in portfolio_list]   # :75-78 ('' dropped)
        self.portfolio_list = [[item + upper_u + 1 for item in sub] for sub in pl]     # :80
        if seed is not None:
            self.seed = seed
            self.random_state = np.random.RandomState(self.seed)

    def sample(self, size, available_log=None):
        out = []
        for i, _ in enumerate(self.src_list):
            available = np.setdiff1d(self.dst_unique, self.portfolio_list[i])      # :96
            if available_log is not None:
                available_log.append(available)
            replace = len(available) < size                                        # :99
            rng = self.random_state if self.seed is not None else np.random
            out.append(rng.choice(available, size=size, replace=replace))          # :101-111
        return np.array(out)
