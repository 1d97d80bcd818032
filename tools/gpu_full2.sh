#!/bin/bash
# full GPU suite (+ the per-element figures of the C2 weight-gradient test), smoke, default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-full}; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { echo "pytest rc=$rc"; grep -n "Error\|assert\|FAILED" $out/pytest.log | head -30; exit $rc; }
timeout -k 10 300 python -m pytest "tests/test_gpu_full_size.py::test_full_size_weight_gradients_per_element_against_fp64_of_the_same_operands" -m gpu -q -s 2>&1 | grep "per-element" | tee $out/per_element.txt
timeout -k 10 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 3; }
tail -1 $out/smoke.log
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 4; }
tail -1 $out/bench.json | cut -c1-600
