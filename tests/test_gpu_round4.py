"""Round-4 GPU tests: the bench's rank path on RCCL at world 1, (later sections) oracle parity at the benched dropout and
per-element evidence for the fp16-split contractions."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]

if has_gpu():
    import pfotgnrec_amd as P
    from pfotgnrec_amd import _lib
    DEV = torch.device("cuda:0")


# ------------------------------------------------------------------ bench.py as a rank
@pytest.mark.parametrize("allreduce", ["single", "buckets"])
def test_bench_rank_path_on_rccl_world1(allreduce):
    """`bench.py --gpus 1` under a torchrun-style environment (RANK=0, WORLD_SIZE=1, MASTER_*) with PFO_DIST_FORCE=1: the
    process group is initialised on RCCL (device bound, explicit timeout), every step all-reduces the real flat gradient
    buffer through ncclAllReduce (one piece / two buckets with the backward's event), the timed blocks are bracketed by
    barriers and a MAX all-reduce, and the line carries collective_ms_per_step / compute_ms_per_step and the secondary run
    with the other all-reduce form.  A child process with a hard limit: a hang cannot take the suite down."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29300 + (os.getpid() % 200) + (7 if allreduce == "buckets" else 0)), PFO_DIST_FORCE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("PFO_DIST_BACKEND", None)
    out = subprocess.run(["timeout", "-k", "10", "420", sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--config", "C2",
                          "--batch", "64", "--steps", "8", "--warmup", "3", "--min-seconds", "0.2", "--no-cpu-baseline", "--graph", "off",
                          "--allreduce", allreduce], env=env, capture_output=True)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    rec = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])
    cfg = rec["config"]
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert cfg["collective"].startswith("rccl all-reduce"), cfg["collective"]
    assert cfg["allreduce"] == allreduce
    assert cfg["collective_samples"] >= 1 and 0 < cfg["collective_ms_per_step"] < rec["ms_per_step"]
    assert abs(cfg["compute_ms_per_step"] + cfg["collective_ms_per_step"] - rec["ms_per_step"]) < 1e-3
    other = "buckets" if allreduce == "single" else "single"
    sec = rec["secondary"]
    assert "error" not in sec, sec
    assert sec["allreduce_" + other]["value"] > 0 and sec["allreduce_" + other]["collective_ms_per_step"] > 0
