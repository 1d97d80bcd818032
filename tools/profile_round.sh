#!/bin/bash
# Round profile of ONE workload on the GPU box: kernel trace + stats of the bench, then the HBM and SQ counter passes (separate
# --pmc runs, as MI355X_MICROARCH.md prescribes; GPU_MAX_HW_QUEUES is exported here because the profiler's preloaded library
# initialises the GPU before Python starts - the package's import-time default comes too late under rocprofv3).
# Usage: tools/profile_round.sh <tag> <key> [bench args]     e.g.  tools/profile_round.sh r5 C3@512 --config C3
# Outputs land in gpurun_out/<tag>_<key>/; tools/summarize_profile.py <tag> <key> turns them into
# profiles/<tag>_kernel_stats_bench_<key>.csv and profiles/<tag>_summary_<key>.json (bench.py reads those by key).
tag=${1:-r6}; key=${2:-C2@512}; shift $(( $# < 2 ? $# : 2 ))
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
out=gpurun_out/${tag}_${key}
mkdir -p $out
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --no-drop-in --min-seconds 0 $*"
# the bench line of the SAME build on the same box (no tracer), embedded in the summary (bench_line_same_build)
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drop-in $* 2>/dev/null | tail -1 > $out/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $CMD > $out/bench_under_trace.log 2>&1
# one step kernel by kernel (before the raw per-dispatch table is dropped below)
mkdir -p gpurun_out/profiles_out
python3 tools/step_timeline.py "$(find $out/trace -name '*kernel_trace.csv' | head -1)" > gpurun_out/profiles_out/${tag}_step_timeline_${key}.txt 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- $CMD > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o p -- $CMD > $out/pmc_write.log 2>&1
# SQ instruction counts of the same command (their own passes): the vector-issue roofline of the attention kernels and the
# matrix-pipe occupancy of the contractions are priced from these (bench.py family_roofline)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/pmc_sq1 -o p -- $CMD > $out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_sq2 -o p -- $CMD > $out/pmc_sq2.log 2>&1
python3 tools/summarize_profile.py $tag $key 25 "$*" > $out/summary.txt 2>&1
mkdir -p gpurun_out/profiles_out && cp profiles/${tag}_kernel_stats_bench_${key}.csv profiles/${tag}_summary_${key}.json gpurun_out/profiles_out/   # (only gpurun_out/ travels back)
# keep what the summary needs, drop the raw per-dispatch tables (tens of MB)
find $out -name '*counter_collection.csv' -delete; find $out -name '*kernel_trace.csv' -delete; find $out -name '*agent_info.csv' -delete
head -14 $out/summary.txt
