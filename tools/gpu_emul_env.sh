#!/bin/bash
# emulated rank-0 step (weak scaling, C2, 512 per GPU) at world 1 / 8 under environment settings: VARIANTS="A=1 B=2"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in ${VARIANTS:-"X=1"}; do
  echo "== $v"
  env ${v//,/ } python bench.py --emulate-ranks ${WORLDS:-1,8} --scaling weak --steps 20 --warmup 5 --min-seconds 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print([(r['world'], r['ms_per_step']) for r in d['table']])"
done; done
