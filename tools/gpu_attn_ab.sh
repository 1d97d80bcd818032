#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/attn_ab; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_tgn_step.py tests/test_gpu_full_size.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $out/pytest.log | head -20; exit $rc; }
L=$GRAFT_REPO_ROOT/pfotgnrec_amd/lib
for v in "X=1" "PFOTGN_LIB=$L/libpfotgn_rc6.so" "PFOTGN_LIB=$L/libpfotgn_rc8.so" "X=1"; do
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 0.5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['block_ms_per_step'], d['roofline']['families_ms_per_step'])"
done 2>&1 | tee $out/ab.txt
