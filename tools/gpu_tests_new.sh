#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/newtests; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k "kink_masked_gradients_of_a_64 or evaluation_batch_slice" > $out/pytest.log 2>&1; rc=$?
tail -4 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $out/pytest.log | head -30; }
timeout -k 10 300 python tools/bench_eval.py > $out/eval.json 2> $out/eval.err; tail -1 $out/eval.json
