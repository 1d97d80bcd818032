#!/bin/bash
# like gpu_ab_lib.sh but prints every kernel family's ms/step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_lib_fam; mkdir -p $out
for rep in 1 2; do
for t in ${LIBS:-"-"}; do
  lib=pfotgnrec_amd/lib/libpfotgn.so; [ "$t" != "-" ] && lib=pfotgnrec_amd/lib/libpfotgn_$t.so
  echo "== $t"
  PFOTGN_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 1.0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step']['median'], d['roofline']['families_ms_per_step'])"
done; done 2>&1 | tee $out/ab.txt
