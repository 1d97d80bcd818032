#!/bin/bash
# evaluation batch (tools/bench_eval.py) under environment variants, one box.  Usage: VARIANTS="A=1 B=2" tools/gpu_eval_ab.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/eval_ab; mkdir -p $out; : > $out/ab.txt
for v in ${VARIANTS:-"X=1"}; do
  echo "== $v" | tee -a $out/ab.txt
  env $v python tools/bench_eval.py 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_batch'], d['ms_all'], d['recall_at_5'])" | tee -a $out/ab.txt
done
