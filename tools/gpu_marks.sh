#!/bin/bash
# untraced critical-path segments of the default step (library milestones).  Usage: tools/gpu_marks.sh [env assignments...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/marks; mkdir -p $out
env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 0.5 --marks 60 ${BENCH_ARGS} > $out/line.json 2> $out/marks.txt
grep -A 120 "^milestones" $out/marks.txt
python -c "
import json;d=json.loads(open('$out/line.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])"
