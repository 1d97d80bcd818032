#!/bin/bash
# kernel + step-level parity tests (everything but the full-size file)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/core; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tgn_step.py tests/test_gpu_round3.py tests/test_gpu_data_parallel.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $out/pytest.log | head -30; exit $rc; }
exit 0
