#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>_<key>/ (tools/profile_round.sh <tag> <key> ...) into profiles/<tag>_kernel_stats_bench_<key>.csv
and profiles/<tag>_summary_<key>.json (key = configuration@batch per GPU, e.g. C2@512 - bench.py looks its own key up): per-kernel launches / average duration from the kernel trace and HBM bytes per launch
from the FETCH_SIZE / WRITE_SIZE passes (separate runs: they do not fit one pass) with the gfx950 correction of
MI355X_MICROARCH.md: the counters are in KiB, and FETCH_SIZE reports half of the bytes of wide coalesced reads
(128-byte requests tallied at 64 B), so HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv, glob, json, os, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
key = sys.argv[2] if len(sys.argv) > 2 else "C2@512"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
bench_args = sys.argv[4] if len(sys.argv) > 4 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "%s_%s" % (tag, key))


def find(sub, pat):
    hits = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0].strip()


trace = find("trace", "*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
# steps in the trace = launches of a once-per-step kernel (the BPR loss): configurations whose bench first times the HIP-graph form
# against the eager one (batch <= 256) run more steps than --steps + --warmup
if "bpr_kernel" in dur:
    steps = len(dur["bpr_kernel"])
kern = {k: {"launches": len(v), "avg_us": round(sum(v) / len(v), 2), "ms_per_step": round(sum(v) / steps / 1e3, 4)}
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))}
stats = find("trace", "*kernel_stats.csv")
if stats:
    dst = os.path.join(root, "profiles", "%s_kernel_stats_bench_%s.csv" % (tag, key))
    open(dst, "w").write(open(stats).read())


def counter(sub, name):
    f = find(sub, "*counter_collection.csv")
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = short(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}


fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
pmc = {}
for k in fetch:
    f, w = fetch[k], write.get(k, 0.0)
    pmc[k] = {"FETCH_SIZE_per_launch": round(f, 1), "WRITE_SIZE_per_launch": round(w, 1),
              "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
for sub, names in (("pmc_sq1", ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")),
                   ("pmc_sq2", ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"))):
    for name in names:
        for k, v in counter(sub, name).items():
            if k in pmc:
                pmc[k].setdefault("sq", {})[name] = round(v, 1)
bench_line = None
try:
    bl = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
    bench_line = {"value": bl["value"], "unit": bl["unit"], "ms_per_step": bl["ms_per_step"], "timed_steps": bl["config"].get("timed_steps"),
                  "block_ms_per_step": bl["config"].get("block_ms_per_step"),
                  "note": "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drop-in %s on the box and build of these passes, no tracer" % bench_args}
except Exception:
    pass
out = {"workload": key,
       "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 "
                  "--no-cpu-baseline --no-prof --min-seconds 0 %s  (+ separate --pmc passes: FETCH_SIZE; WRITE_SIZE; "
                  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS; SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; "
                  "GPU_MAX_HW_QUEUES=16 exported)" % bench_args,
       "steps_in_trace": steps, "kernels": kern,
       "pmc": {"correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes (gfx950, MI355X_MICROARCH.md)", "kernels": pmc},
       "bench_line_same_build": bench_line}
json.dump(out, open(os.path.join(root, "profiles", "%s_summary_%s.json" % (tag, key)), "w"), indent=1)
top = list(kern.items())[:12]
for k, v in top:
    print("%-50s n=%5d avg %8.1f us  %7.4f ms/step  hbm/launch %s" % (k[:50], v["launches"], v["avg_us"], v["ms_per_step"],
          pmc.get(k, {}).get("hbm_bytes_per_launch_corrected")))
