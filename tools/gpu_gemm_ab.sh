#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gemm_ab
for v in ${VARIANTS:-"PFO_GEMM_AREG=1" "PFO_GEMM_AREG=3" "PFO_GEMM_AREG=1" "PFO_GEMM_AREG=3"}; do
  echo "== $v"
  env BX_API=1 $v timeout -k 10 200 python tools/bench_gemm_bf16x3.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm_ab/log.txt
done
