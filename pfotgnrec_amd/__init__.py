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
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"
    # The assignment only takes effect if the HIP runtime has not been initialised yet in this process: torch.cuda was used
    # before the import, or the process runs under `rocprofv3 --pmc ...` whose preloaded library initialises the GPU before
    # Python starts (tools/profile_round.sh therefore exports the variable itself).  Say so instead of silently running on
    # the default 4 queues (the side streams sit in priority classes of their own, so the loss is small on one GPU; on the
    # rank path with c10d / RCCL streams it is not).
    try:
        import sys as _sys
        _t = _sys.modules.get("torch")
        if _t is not None and _t.cuda.is_initialized():
            import warnings as _warnings
            _warnings.warn("pfotgnrec_amd: the GPU was initialised before this import, so GPU_MAX_HW_QUEUES=16 cannot take effect; "
                           "export GPU_MAX_HW_QUEUES=16 in the environment (INTEGRATION.md)", RuntimeWarning, stacklevel=2)
    except Exception:
        pass

from .data import Data, compute_time_statistics, get_data  # noqa: F401
from .neighbor_finder import NeighborFinder, get_neighbor_finder  # noqa: F401
from .rand_edge_sampler import RandEdgeSampler, DeviceNegativeSampler  # noqa: F401
from .mv_sampler import MVSampler  # noqa: F401
from .memory import Memory  # noqa: F401
from .tgn import TGN  # noqa: F401
from .optim import FusedAdam, overlap_backward  # noqa: F401
from .functional import bpr_loss, bpr_loss_blocks, bpr_step, time_encode, rank_metrics  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401,E402
from . import ops  # noqa: F401,E402  (registers torch.ops.pfotgn.*)
