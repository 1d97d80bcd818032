#!/bin/bash
# Round-4 profile set, part 2: one line per configuration (+ evaluation batch), emulated-rank tables.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
bash tools/bench_all_configs.sh gpurun_out/r4/bench_all_configs.jsonl
python tools/bench_eval.py 2>/dev/null | tail -1 > gpurun_out/r4/eval_batch.json
: > gpurun_out/r4/emulated_ranks.jsonl
timeout -k 10 400 python bench.py --emulate-ranks 1,2,4,8 --scaling weak --steps 20 --warmup 5 --min-seconds 1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r4/emulated_ranks.jsonl
timeout -k 10 600 python bench.py --emulate-ranks 1,2,4,8 --steps 20 --warmup 5 --min-seconds 1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r4/emulated_ranks.jsonl
wc -l gpurun_out/r4/emulated_ranks.jsonl
