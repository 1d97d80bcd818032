#!/bin/bash
# A/B on ONE box between bench.py argument sets: ARGS_A="..." ARGS_B="..." (interleaved, 3 rounds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_args; mkdir -p $out
for rep in 1 2 3; do
for v in A B; do
  if [ $v = A ]; then a="$ARGS_A"; else a="$ARGS_B"; fi
  echo "== $v: $a"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --min-seconds 1.0 $a 2>$out/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step'], d['config'].get('final_loss'))" || tail -5 $out/err.txt
done; done 2>&1 | tee $out/ab.txt
