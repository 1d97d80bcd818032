cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/c3tr; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3tr -o t -- python3 bench.py --config C3 --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 0 > /dev/null 2>&1
f=$(find gpurun_out/c3tr -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:40]:
    print("%-60s calls %5s avg %8.1f us total/25 %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/25e3))
PY
find gpurun_out/c3tr -type f ! -name '*kernel_stats.csv' -delete
