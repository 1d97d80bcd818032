#!/bin/bash
# A/B on ONE box between library builds: LIBS="<tag or empty> ..." (pfotgnrec_amd/lib/libpfotgn_<tag>.so; "-" = the default build),
# each run twice, interleaved, with the kernel-family brackets on.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_lib; mkdir -p $out
for rep in 1 2; do
for t in ${LIBS:-"-"}; do
  lib=pfotgnrec_amd/lib/libpfotgn.so; [ "$t" != "-" ] && lib=pfotgnrec_amd/lib/libpfotgn_$t.so
  echo "== $t"
  PFOTGN_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 1.0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d['roofline']['families_ms_per_step']
print(d['ms_per_step'], d['config']['block_ms_per_step']['median'], {k:f[k] for k in ('attn_bwd_runs','attn_fwd','gemm_bx','gemm_tn_bx') if k in f})"
done; done 2>&1 | tee $out/ab.txt
