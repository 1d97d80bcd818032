#!/bin/bash
# A/B on ONE box between environment settings: VARIANTS="A=1 B=2 ..." (each run twice, interleaved)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_env; mkdir -p $out
for rep in 1 2; do
for v in ${VARIANTS:-"X=1"}; do
  echo "== $v"
  env ${v//,/ } python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 1.0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step'])"
done; done 2>&1 | tee $out/ab.txt
