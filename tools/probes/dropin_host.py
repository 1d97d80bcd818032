"""The drop-in loops of bench.py (secondary.drop_in_surface) three times over in ONE process, then once more with perf_counter
wrappers around the package's own host functions (host speed differs from box to box: only same-call comparisons mean anything)."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-prof"]
args = bench.parse()
wl = bench.Workload(args, "C2", torch.device("cuda:0"), 0, 1, "weak")
for i in range(5):
    wl.step(i)
torch.cuda.synchronize()
for rep in range(3):
    out = bench.drop_in_surface(wl, budget_s=0.5)
    for k in bench.DROP_IN_ENTRIES:
        print("%-32s %.4f ms  host %s" % (k, out[k]["ms_per_step"], [round(x, 3) for x in out[k]["host_ms_per_step"].values()]), flush=True)

# host time inside the package's own functions (perf_counter around each, per step)
import time, collections
acc, cnt = collections.Counter(), collections.Counter()
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
            cnt[label] += 1
    setattr(obj, name, g)
import pfotgnrec_amd as P
from pfotgnrec_amd import rand_edge_sampler as RS, _lib
tgn = wl.tgn
for n in ("_native_forward", "_native_backward", "_batch_to_dev", "_make_call", "embed_device", "_attach_grads", "_state_struct",
          "_assemble_roots", "_take_prefetched", "_check_nodes", "_check_edges", "hot_parameters", "_torch_versions", "_param_key"):
    wrap(tgn, n)
wrap(_lib, "call", "_lib.call")
wrap(RS, "packed_portfolios_of")
wrap(RS, "_device_rows")
wrap(RS.DeviceNegativeSampler, "sample", "neg.sample(dev)")
wrap(torch, "cat", "torch.cat")
out = bench.drop_in_surface(wl, budget_s=0.5)
steps = sum(out[k]["timed_steps"] + 3 for k in bench.DROP_IN_ENTRIES)
print("wrapped run: FusedAdam %.4f ms" % out["FusedAdam"]["ms_per_step"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-24s %7.1f us per step   (%.1f calls per step)" % (k, 1e6 * v / steps, cnt[k] / steps))
