#!/bin/bash
# Round-4 profile set, part 1: kernel trace + stats, HBM and SQ counter passes of the default bench, the bench line itself,
# the untraced milestones, one traced step's timeline.  (tools/summarize_profile.py r4 condenses it into profiles/.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r4 > gpurun_out/profile_round_r4.log 2>&1
tail -2 gpurun_out/profile_round_r4.log | cut -c1-300
f=$(find gpurun_out/r4/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f > gpurun_out/r4/timeline.txt 2>&1
tail -1 gpurun_out/r4/timeline.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 0.5 --marks 100 > gpurun_out/r4/marks_line.json 2> gpurun_out/r4/marks.txt
grep -c "us  n=" gpurun_out/r4/marks.txt
bash tools/pmc_kernel.sh attn_ gpurun_out/pmc_attn > gpurun_out/r4/pmc_sq_counters_attn.txt 2>&1
bash tools/pmc_kernel.sh gemm_ gpurun_out/pmc_gemm > gpurun_out/r4/pmc_sq_counters_gemm.txt 2>&1
head -5 gpurun_out/r4/pmc_sq_counters_gemm.txt
