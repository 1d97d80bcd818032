#!/bin/bash
# emulated-rank table (weak, C2) with and without the side-stream tail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
for t in 1 0; do
  python bench.py --emulate-ranks 1,2,4,8 --scaling weak --steps 20 --warmup 5 --min-seconds 1 --no-cpu-baseline --overlap-tail $t 2>gpurun_out/r4/emul_err.txt | tail -1 > gpurun_out/r4/emulated_weak_t$t.json
  python -c "
import json
d=json.load(open('gpurun_out/r4/emulated_weak_t$t.json')); print('overlap-tail $t', [(r['world'], r['ms_per_step']) for r in d['table']])" || tail -5 gpurun_out/r4/emul_err.txt
done
