#!/bin/bash
# full GPU suite + the default bench line + a kernel trace / timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-full}; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc"; grep -n "Error\|assert\|FAILED" $out/pytest.log | head -30; exit $rc; }
timeout -k 10 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 3; }
tail -1 $out/smoke.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 4; }
tail -1 $out/bench.json | cut -c1-2600
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 0 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 5; }
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f > $out/timeline.txt 2>&1
tail -1 $out/timeline.txt
