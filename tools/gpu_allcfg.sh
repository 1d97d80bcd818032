#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/allcfg
timeout -k 10 240 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k deterministic 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 1.0 --config C1 --graph off 2>/dev/null | tail -1 > gpurun_out/allcfg/c1_eager.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 1.0 --config C2 --graph on 2>/dev/null | tail -1 > gpurun_out/allcfg/c2_graph.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 1.0 --config C2 --batch 128 --graph off 2>/dev/null | tail -1 > gpurun_out/allcfg/c2_128_eager.json
for f in c1_eager c2_graph c2_128_eager; do python -c "
import json,sys; d=json.load(open('gpurun_out/allcfg/$f.json')); print('$f', d['ms_per_step'], d['config']['hip_graph'], d['config']['host_enqueue_ms_per_step'])"; done
bash tools/bench_all_configs.sh gpurun_out/allcfg/all.jsonl
