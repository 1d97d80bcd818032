#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bis
for v in "X=0" "PFO_DBG_JOIN=1" "PFO_DBG_MARK=1" "PFO_DBG_GRAD=1" "PFO_DBG_MEAN=1" "PFO_DBG_JOIN=1 PFO_DBG_MARK=1 PFO_DBG_GRAD=1 PFO_DBG_MEAN=1"; do
  env $v timeout -k 10 200 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "graphed and False" > gpurun_out/bis/log_$(echo $v | tr ' =' '__').txt 2>&1
  echo "== $v rc=$?"
done
