#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r3 > gpurun_out/profile_round_r3.log 2>&1
tail -2 gpurun_out/profile_round_r3.log | cut -c1-300
bash tools/pmc_kernel.sh gemm_bx_areg gpurun_out/pmc_gemm > gpurun_out/r3_pmc_sq_counters_gemm.txt 2>&1
bash tools/pmc_kernel.sh attn_bwd_runs gpurun_out/pmc_attn > gpurun_out/r3_pmc_sq_counters_attn.txt 2>&1
head -24 gpurun_out/r3_pmc_sq_counters_attn.txt
