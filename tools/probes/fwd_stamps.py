"""Where a wavefront of the attention forward (attn_fwd_ring_kernel) spends its cycles (GPU box; diagnostic build):
    SRC=attn tools/probes/mkvariant.sh fstamps -DFWD_STAMPS=1
    PFOTGN_LIB=$PWD/pfotgnrec_amd/lib/libpfotgn_fstamps.so python tools/probes/fwd_stamps.py
Runs the default bench workload (C2) for a few steps and prints the s_memtime sums per section (both layers' launches together;
layer 1 holds 95 % of the instances).  The stamped build runs slower than the product build: proportions only."""
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
out = (C.c_ulonglong * 16)()
assert lib.pfo_attn_fwd_stamps(out, 1) == 0
steps = 20
for i in range(10, 10 + steps):
    w.step(i)
torch.cuda.synchronize()
assert lib.pfo_attn_fwd_stamps(out, 0) == 0
v = [out[i] / steps for i in range(10)]
tot, waves, pairs = v[0], v[7], v[8]
print("per step: wavefront cycles %.3e  wavefronts with a neighbour %.0f  pairs %.0f (%.2f per wavefront)" % (tot, waves, pairs, pairs / max(waves, 1)))
names = ["", "prologue up to the barrier (first-level loads)", "query row + first DMAs issued", "wait for the pair's DMA",
         "LDS reads + time encoding + scores + reduce", "softmax + context update", "epilogue (stores)"]
acc = 0.0
for i in range(1, 7):
    acc += v[i]
    per = v[i] / max(pairs if i in (3, 4, 5) else waves, 1)
    print("  %-48s %.3e cycles  %5.1f %%   %8.0f cycles per %s" % (names[i], v[i], 100.0 * v[i] / tot, per, "pair" if i in (3, 4, 5) else "wavefront"))
print("  %-48s %.3e cycles  %5.1f %%" % ("rest", tot - acc, 100.0 * (tot - acc) / tot))
