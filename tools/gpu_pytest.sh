#!/bin/bash
# a pytest selection on the GPU box.  Usage: tools/gpu_pytest.sh <outdir-name> <pytest args...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1; shift; mkdir -p $out
timeout -k 10 1000 python -m pytest "$@" -m gpu -q > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert \|FAILED" $out/pytest.log | head -40; }
exit $rc
