#!/bin/bash
# Round profile on the GPU box: kernel trace + stats of the default bench, then the two HBM counter passes
# (separate --pmc runs, as MI355X_MICROARCH.md prescribes).  Usage: tools/profile_round.sh r1
# Outputs land in gpurun_out/<tag>/; tools/summarize_profile.py turns them into profiles/<tag>_*.
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $CMD > $out/bench_under_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- $CMD > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o p -- $CMD > $out/pmc_write.log 2>&1
# SQ instruction counts of the same command (their own passes): the vector-issue roofline of the attention kernels and the
# matrix-pipe occupancy of the contractions are priced from these (bench.py family_roofline)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/pmc_sq1 -o p -- $CMD > $out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_sq2 -o p -- $CMD > $out/pmc_sq2.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-400
