import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-prof"]
args = bench.parse()
wl = bench.Workload(args, "C2", torch.device("cuda:0"), 0, 1, "weak")
for i in range(5): wl.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
out = bench.drop_in_surface(wl, budget_s=0.6)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
print({k: v["ms_per_step"] for k, v in out.items() if isinstance(v, dict)})
