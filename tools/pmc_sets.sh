#!/bin/bash
# counter passes of the caller's choosing for kernels of the bench step (GPU box; kernels run one at a time under --pmc, so
# GRBM_GUI_ACTIVE is a kernel's duration ALONE).  Usage: SETS="A B;C" tools/pmc_sets.sh <kernel-substring[,more]> [outdir]
# The wait / back-pressure set of rounds 4-5 (profiles/r*_pmc_wait_counters_attn.txt):
#   SETS="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_IFETCH;SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH;SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32;SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM;SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU"
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}     # (the profiler initialises the GPU before Python: the package's default comes too late)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
kern=$1
out=${2:-gpurun_out/pmc_sets}
CMD="python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-prof --no-drop-in --min-seconds 0 ${BENCH_ARGS}"
IFS=';' read -ra sets <<< "${SETS:-FETCH_SIZE;WRITE_SIZE;GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES}"
for set in "${sets[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o p -- $CMD > /dev/null 2>&1
done
python3 - "$out" "$kern" <<'PY'
import sys, glob, csv, collections
out, kerns = sys.argv[1], sys.argv[2].split(",")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(x in k for x in kerns): continue
        key = (k.split("(")[0][:44], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
for key, d in sorted(agg.items()):
    print(key)
    for c, v in sorted(d.items()):
        print("   %-28s %.5g per launch" % (c, v / max(1, cnt[(key, c)])))
PY
find $out -type f ! -name '*.txt' -delete
