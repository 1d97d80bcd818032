#!/bin/bash
# per-kernel times of tools/bench_gemm_bf16x3.py (image kernel and contraction kernel separately)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gemm_trace; mkdir -p $out
export BX_API=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- python3 tools/bench_gemm_bf16x3.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 5; }
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
seq = [(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X","")) for r in rows]
d = collections.OrderedDict()
for n, t, g in seq:
    if "bimg" in n or "gemm_bx" in n:
        d.setdefault((n, g), []).append(t)
for (n, g), v in d.items():
    v = sorted(v); print("%-62s grid %8s  n=%3d  median %7.1f us" % (n, g, len(v), v[len(v)//2]))
P
