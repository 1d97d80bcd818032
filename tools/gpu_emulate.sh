#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/emul; mkdir -p $out
timeout -k 10 600 python bench.py --emulate-ranks 1,2,4,8 --scaling weak --steps 20 --warmup 5 --min-seconds 1 --no-cpu-baseline > $out/emulated_C2_weak.json 2> $out/err.txt || { tail -20 $out/err.txt; exit 1; }
cat $out/emulated_C2_weak.json
timeout -k 10 900 python bench.py --emulate-ranks 1,2,4,8 --steps 20 --warmup 5 --min-seconds 1 --no-cpu-baseline > $out/emulated_C4.json 2>> $out/err.txt || { tail -20 $out/err.txt; exit 1; }
cat $out/emulated_C4.json
