#!/bin/bash
# SQ/LDS counter passes for one kernel of the bench step (GPU box).  Usage: tools/pmc_kernel.sh <kernel-substring> [outdir]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
kern=$1
out=${2:-gpurun_out/pmc_kernel}
CMD="python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-prof --min-seconds 0"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT"; do
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
        print("   %-28s %.4g per launch" % (c, v / max(1, cnt[(key, c)])))
PY
