#!/bin/bash
# step-level parity tests, then the A/B against libpfotgn_prev.so on the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_prev; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_tgn_step.py tests/test_gpu_round3.py tests/test_gpu_data_parallel.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $out/pytest.log | head -20; exit $rc; }
bash tools/gpu_ab_prev.sh
