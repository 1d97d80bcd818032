#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/attn_abl; mkdir -p $out
for v in "PFO_ATTN_ABL=0" "PFO_ATTN_ABL=2" "PFO_ATTN_ABL=3" "PFO_ATTN_ABL=4" "PFO_ATTN_ABL=5" "PFO_ATTN_ABL=0"; do
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 0.3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['families_ms_per_step']['attn_bwd_runs'])"
done 2>&1 | tee $out/abl.txt
