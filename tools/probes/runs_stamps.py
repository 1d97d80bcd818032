"""Where a wavefront of the layer-1 attention backward spends its cycles (GPU box; diagnostic build of the library):
    python -m pfotgnrec_amd.build -DRUNS_STAMPS=1 --tag=stamps
    PFOTGN_LIB=$PWD/pfotgnrec_amd/lib/libpfotgn_stamps.so python tools/probes/runs_stamps.py
Runs the default bench workload (C2) for a few steps and prints the s_memtime sums per section of attn_bwd_runs_kernel."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402
from pfotgnrec_amd import _lib  # noqa: E402

sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-prof"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda:0")
w = bench.Workload(args, args.config or "C2", dev, 0, 1, "weak")
for i in range(10):
    w.step(i)
torch.cuda.synchronize()
lib = _lib.load()
out = (C.c_ulonglong * 8)()
assert lib.pfo_attn_runs_stamps(out, 1) == 0
steps = 20
for i in range(10, 10 + steps):
    w.step(i)
torch.cuda.synchronize()
assert lib.pfo_attn_runs_stamps(out, 0) == 0
tot, setup, walk, flush, store, members, chunks = [out[i] / steps for i in range(7)]
print("per launch: wavefront cycles %.3e  members %.0f  chunks %.0f" % (tot, members, chunks))
for name, v in (("member set-up", setup), ("key walk", walk), ("flush (rows -> atomics)", flush), ("row-sum store", store),
                ("chunk prologue / epilogue / rest", tot - setup - walk - flush - store)):
    print("  %-34s %.3e cycles  %5.1f %%   %8.0f cycles per member" % (name, v, 100.0 * v / tot, v / max(members, 1)))
