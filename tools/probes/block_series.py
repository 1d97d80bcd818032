"""Per-block step times of the default bench loop (20-step blocks over ~2 s), with the cycle collector on and off: is the
first -> last drift of config.block_ms_per_step the workload's (later batches touch more nodes with pending messages) or the host's?"""
import gc, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-prof", "--no-drop-in"]
args = bench.parse()
wl = bench.Workload(args, "C2", torch.device("cuda:0"), 0, 1, "weak")
for mode in ("gc on", "gc off", "gc on", "gc off"):
    gc.collect()
    if mode == "gc off":
        gc.disable()
    el, n, blocks, _, _ = wl.timed(20, 5, 2.0, 0)
    gc.enable()
    b = wl.block_ms
    print("%-7s %.4f ms per step over %d blocks; every 8th block: %s" % (mode, 1e3 * el / n, blocks, [round(x, 3) for x in b[::8]]), flush=True)
