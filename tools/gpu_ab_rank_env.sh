#!/bin/bash
# A/B of environment settings on the RANK path at world 1 (RCCL, PFO_DIST_FORCE=1): VARIANTS="A=1 B=2,C=3 ..." (each twice, interleaved)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_rank_env; mkdir -p $out
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 PFO_DIST_FORCE=1
p=29410
for rep in 1 2; do
for v in ${VARIANTS:-"X=1"}; do
  p=$((p+1))
  echo "== $v"
  env ${v//,/ } MASTER_PORT=$p python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 1.0 --no-secondary ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config'].get('collective_ms_per_step'), d['config']['block_ms_per_step'])"
done; done 2>&1 | tee $out/ab.txt
