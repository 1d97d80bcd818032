#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/det; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_tgn_step.py tests/test_gpu_round2.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $out/pytest.log | head -30; exit $rc; }
for v in "" "--deterministic" "" "--deterministic"; do
  echo "== $v"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 0.5 $v 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['families_ms_per_step'])"
done 2>&1 | tee $out/ab.txt
