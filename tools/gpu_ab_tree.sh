#!/bin/bash
# A/B on ONE box between this tree and another checkout of the repo under ./_r3 (whole tree: library AND host side)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_tree; mkdir -p $out
for rep in 1 2 3; do
for t in . _r3; do
  echo "== $t ${BENCH_ARGS}"
  (cd $t && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 1.0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step'])")
done; done 2>&1 | tee $out/ab.txt
