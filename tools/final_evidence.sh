cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout -k 10 500 python bench.py --steps 20 --warmup 5 2>gpurun_out/final/bench_default.err | tail -1 > gpurun_out/final/r6_bench_default_line.json
tools/bench_all_configs.sh gpurun_out/final/r6_bench_all_configs.jsonl > gpurun_out/final/bench_all.txt 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --no-drop-in --min-seconds 1.0 --marks 100 > gpurun_out/final/marks.txt 2>&1
timeout -k 10 300 python bench.py --emulate-ranks 1,2,4,8 --scaling weak --allreduce fused --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final/r6_emulated_ranks.jsonl
timeout -k 10 300 python bench.py --emulate-ranks 1,2,4,8 --scaling strong --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/final/r6_emulated_ranks.jsonl
PFOTGN_LIB=$GRAFT_REPO_ROOT/pfotgnrec_amd/lib/libpfotgn_fstamps.so timeout -k 10 300 python tools/probes/fwd_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/r6_fwd_stamps.txt
PFOTGN_LIB=$GRAFT_REPO_ROOT/pfotgnrec_amd/lib/libpfotgn_stamps.so timeout -k 10 300 python tools/probes/runs_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/r6_runs_stamps.txt
SETS="SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_IFETCH;SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS;GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SMEM" tools/pmc_sets.sh attn_fwd_ring_kernel,attn_bwd_runs_kernel,attn_bwd_ring_kernel gpurun_out/final/pmc_attn > gpurun_out/final/r6_pmc_wait_counters_attn.txt 2>&1
tail -3 gpurun_out/final/bench_all.txt
