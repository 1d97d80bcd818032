#!/bin/bash
# One driver-format bench line per BASELINE.json configuration (+ C2 at batch 128 and the evaluation batch) for profiles/.
# Usage (GPU box): tools/bench_all_configs.sh <out.jsonl>
out=${1:-gpurun_out/bench_all.jsonl}
: > $out
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drop-in --min-seconds 1.0 "$@" 2>/dev/null | tail -1 >> $out; }
run --config C1
run --config C2
run --config C2 --batch 128
run --config C2 --batch 2048
run --config C3
run --config C4
run --config C5
python tools/bench_eval.py 2>/dev/null | tail -1 >> $out
python - "$out" <<'PY'
import sys, json
for line in open(sys.argv[1]):
    d = json.loads(line)
    if "metric" in d:
        print("%-110s %10.1f /s  %8.4f ms/step" % (d["config"]["workload"][:110], d["value"], d["ms_per_step"]))
    else:
        print("eval batch: %.2f ms (%d roots)" % (d["ms_per_batch"], d["roots"]))
PY
