#!/bin/bash
# A/B of environment switches on the default bench: tools/ab_env.sh "VAR=1" "VAR=2 OTHER=3" ...   ("" = defaults)
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 0.5 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step'], d['roofline']['families_ms_per_step'])"
done
