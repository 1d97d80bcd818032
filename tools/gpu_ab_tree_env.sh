#!/bin/bash
# this tree under several environment settings against the ./_r3 checkout, one box, interleaved
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_tree_env; mkdir -p $out
run() { (cd $1 && env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 1.0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step']['median'], d['config']['host_enqueue_ms_per_step'])"); }
for rep in 1 2; do
  echo "== _r3"; run _r3 X=1
  for v in ${VARIANTS:-"X=1"}; do echo "== . $v"; run . "$(echo $v | tr "," " ")"; done
done 2>&1 | tee $out/ab.txt
