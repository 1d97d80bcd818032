#!/bin/bash
# the N-rank bench path rehearsed with two ranks on the one GPU of the box (gloo between them: RCCL wants one device per rank)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/two; mkdir -p $out
PFO_DIST_BACKEND=gloo PFO_FORCE_DEVICE=0 timeout -k 10 500 python bench.py --gpus 2 --steps 10 --warmup 3 --min-seconds 0.3 --no-cpu-baseline --secondary ${BENCH_ARGS} > $out/line.json 2> $out/err.txt; rc=$?
tail -1 $out/line.json | cut -c1-1500
[ $rc -ne 0 ] && tail -15 $out/err.txt
exit $rc
