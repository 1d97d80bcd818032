#!/bin/bash
# wait / back-pressure counter passes for one kernel of the bench step (GPU box).  Usage: tools/pmc_kernel2.sh <kernel-substring> [outdir]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
kern=$1
out=${2:-gpurun_out/pmc_kernel2}
CMD="python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-prof --min-seconds 0"
for set in "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_IFETCH" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM" "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o p -- $CMD > /dev/null 2>&1
done
python3 - "$out" "$kern" <<'PY'
import sys, glob, csv, collections
out, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kern not in k: continue
        key = (k.split("(")[0][:44], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
for key, d in sorted(agg.items()):
    print(key)
    for c, v in sorted(d.items()):
        print("   %-32s %.4g per launch" % (c, v / max(1, cnt[(key, c)])))
PY
