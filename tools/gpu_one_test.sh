#!/bin/bash
# one test selection on the GPU box: TESTS="tests/x.py -k name"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/one
timeout -k 10 600 python -m pytest ${TESTS} -m gpu -x -q > gpurun_out/one/pytest.log 2>&1; rc=$?
tail -25 gpurun_out/one/pytest.log
exit $rc
