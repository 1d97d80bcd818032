#!/bin/bash
# A/B on ONE box between environment settings with one kernel's average duration from a kernel trace beside the step time:
# VARIANTS="A=1 B=2 ..." KERNEL=<substring of the kernel name>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_env_kern; mkdir -p $out
for v in ${VARIANTS:-"X=1"}; do
  echo "== $v"
  export $v
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 1.0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step'])"
  rm -rf $out/tr_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$v -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof ${BENCH_ARGS} > /dev/null 2>&1
  f=$(find $out/tr_$v -name '*kernel_stats.csv' | head -1)
  python - "$f" "${KERNEL:-segsum}" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in sys.argv[2].split(",")):
        print("   %-44s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  find $out/tr_$v -type f ! -name '*kernel_stats.csv' -delete
  unset ${v%%=*}
done 2>&1 | tee $out/ab.txt
