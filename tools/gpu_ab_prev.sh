#!/bin/bash
# A/B on ONE box: the working tree's library against pfotgnrec_amd/lib/libpfotgn_prev.so (tools: PFO_CSRC=<dir> python -m pfotgnrec_amd.build --tag=prev)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_prev; mkdir -p $out
L=$GRAFT_REPO_ROOT/pfotgnrec_amd/lib
for v in "X=1" "PFOTGN_LIB=$L/libpfotgn_prev.so" "X=1" "PFOTGN_LIB=$L/libpfotgn_prev.so" "X=1" "PFOTGN_LIB=$L/libpfotgn_prev.so"; do
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 1.0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step'])"
done 2>&1 | tee $out/ab.txt
