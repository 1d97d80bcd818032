#!/bin/bash
# PMC passes over the bf16x3 GEMM micro-benchmark (GPU box). Usage: tools/pmc_gemm.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/pmc_gemm}
export PFO_GEMM_BF16X3=2
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o p -- python3 tools/bench_gemm_bf16x3.py > /dev/null 2>&1
done
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/*/*counter_collection.csv") + glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bf16x3" not in k: continue
        key = (k[:40], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
for key, d in agg.items():
    print(key)
    for c, v in sorted(d.items()):
        print("   %-28s %.4g per launch" % (c, v / max(1, cnt[(key, c)])))
PY
