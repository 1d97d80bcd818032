#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-trace}; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 0 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 5; }
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f > $out/timeline.txt 2>&1
head -30 $out/timeline.txt | cut -c1-110; tail -1 $out/timeline.txt
