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
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-400
