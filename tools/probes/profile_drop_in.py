"""cProfile of the drop-in loop's host side (bench.drop_in_surface, one entry), sorted by self time.  ENTRY=<index into
bench.DROP_IN_ENTRIES> (default: FusedAdam(overlap_backward=True))."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
entry = bench.DROP_IN_ENTRIES[int(os.environ.get("ENTRY", "4"))]
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-prof"]
args = bench.parse()
wl = bench.Workload(args, "C2", torch.device("cuda:0"), 0, 1, "weak")
for i in range(5):
    wl.step(i)
torch.cuda.synchronize()
bench.DROP_IN_ENTRIES = (entry,)
bench.drop_in_surface(wl, budget_s=0.3)                # caches warm (packed portfolios, staging ring, optimizer state)
pr = cProfile.Profile()
pr.enable()
out = bench.drop_in_surface(wl, budget_s=1.0)
pr.disable()
n = out[entry]["timed_steps"] + 3
print(entry, out[entry]["ms_per_step"], "ms per step under the profiler,", n, "steps")
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
txt = s.getvalue()
print(txt[txt.index("ncalls"):][:7000])
