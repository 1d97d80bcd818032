#!/bin/bash
# timing-only ablations / A-B builds of gemm_bx_areg_kernel at the h1 / dctx' shapes (GPU box): VARIANTS="abl1 abl2 nt ..." name
# pfotgnrec_amd/lib/libpfotgn_<tag>.so builds (tools/probes/mkvariant.sh <tag> -DBXA_ABL=<bits> | -DBXA_NT=1 | -DBXA_STAGGER=n)
cd "$GRAFT_REPO_ROOT"
for v in "" ${VARIANTS:-abl1 abl2 abl4 abl8 abl3 abl12 abl31}; do
  lib=$GRAFT_REPO_ROOT/pfotgnrec_amd/lib/libpfotgn${v:+_$v}.so
  [ -f $lib ] || continue
  echo "== ${v:-base}"
  PFOTGN_LIB=$lib BX_API=1 python tools/bench_gemm_bf16x3.py 2>/dev/null | grep -v 'PFO_GEMM' | cut -c1-${COLS:-70}
done
