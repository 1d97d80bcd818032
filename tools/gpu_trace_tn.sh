#!/bin/bash
# TN kernel durations in one traced step for the default library and a variant (PFOTGN_LIB)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default "$1"; do
  out=gpurun_out/trtn_$(basename "$v" .so); mkdir -p $out
  if [ "$v" != default ]; then export PFOTGN_LIB="$v"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 0 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 5; }
  f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
  echo "== $v"
  python3 - "$f" <<'P'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "tn_group" in n or "areg" in n or "attn_bwd_runs" in n:
        d[(n[:42], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v = sorted(v); print("%-44s grid %8s n=%3d median %7.1f us" % (k[0], k[1], len(v), v[len(v)//2]))
P
done
