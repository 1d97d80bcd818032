"""MI355X-native TGN-recommender training path (drop-in for youngandbin/PfoTGNRec's hot path).

Public surface mirrors the reference's: ``TGN`` (model/tgn.py), ``NeighborFinder`` /
``get_neighbor_finder`` / ``RandEdgeSampler`` (utils/utils.py), ``Data`` (utils/data.py), ``Memory``
(modules/memory.py), plus the MV sampler that main.py keeps inline (``MVSampler``).
"""
__version__ = "0.1.0"

import os as _os

# The library overlaps three internal side streams with the caller's stream; the HIP runtime maps all streams of a process onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) - with a process group (RCCL + c10d streams) or other extra streams around, a
# side stream ends up behind the caller's stream in ONE hardware queue and the overlap is lost (bench.py, DESIGN 6: +18 % per
# step on the rank path).  The runtime reads the variable when it initialises: import this package (or set the variable)
# before the first use of the GPU.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .data import Data, compute_time_statistics, get_data  # noqa: F401
from .neighbor_finder import NeighborFinder, get_neighbor_finder  # noqa: F401
from .rand_edge_sampler import RandEdgeSampler, DeviceNegativeSampler  # noqa: F401
from .mv_sampler import MVSampler  # noqa: F401
from .memory import Memory  # noqa: F401
from .tgn import TGN  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .functional import bpr_loss, bpr_step, time_encode, rank_metrics  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401,E402
from . import ops  # noqa: F401,E402  (registers torch.ops.pfotgn.*)
