#!/bin/bash
# quick GPU check: the step-level parity tests, the default bench line (no CPU baseline), one step timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-quick}
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_tgn_step.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_data_parallel.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc"; grep -n "Error\|assert\|FAILED" $out/pytest.log | head -30; exit $rc; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 4; }
tail -1 $out/bench.json | cut -c1-1800
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 0 > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 5; }
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f > $out/timeline.txt 2>&1
tail -1 $out/timeline.txt
