#!/bin/bash
# weight-gradient tile ablations (timing only): default build, then -DBX_EXP=1/2/3 builds (first tile only / no MFMA / no LDS refill)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tn_abl; mkdir -p $out; : > $out/abl.txt
for t in "" exp1 exp2 exp3; do
  lib=pfotgnrec_amd/lib/libpfotgn.so; [ -n "$t" ] && lib=pfotgnrec_amd/lib/libpfotgn_$t.so
  echo "== ${t:-default}" | tee -a $out/abl.txt
  PFOTGN_LIB=$GRAFT_REPO_ROOT/$lib python tools/bench_gemm_tn.py 2>&1 | tee -a $out/abl.txt
done
