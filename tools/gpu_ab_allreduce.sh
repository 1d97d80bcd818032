#!/bin/bash
# the three all-reduce forms of the rank path on ONE GPU (RCCL, world 1, PFO_DIST_FORCE=1): step time of the default workload
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_allreduce; mkdir -p $out
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 PFO_DIST_FORCE=1
p=29310
for rep in 1 2; do
for m in single buckets fused fused_buckets; do
  p=$((p+1))
  echo "== $m"
  MASTER_PORT=$p python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 1.0 --no-secondary --allreduce $m ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config'].get('collective_ms_per_step'), d['config']['block_ms_per_step'])"
done; done 2>&1 | tee $out/ab.txt
