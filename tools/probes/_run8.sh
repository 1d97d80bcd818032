cd $GRAFT_REPO_ROOT
tools/gpu.sh test k12 tests/test_gpu_kernels.py tests/test_gpu_round3.py tests/test_gpu_round4.py -k "tn or weight or gemm or grad or gru" || exit 1
for v in 500 0; do echo "== PFO_TN8=$v"; PFO_TN8=$v python tools/bench_gemm_tn.py 2>/dev/null; done
tools/gpu.sh ab ab_tn8 "- PFO_TN8=0 PFO_TN8=600"
