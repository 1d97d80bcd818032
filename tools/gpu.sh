#!/bin/bash
# The one GPU-box runner (replaces the round-3/4 one-off gpu_*.sh scripts).  Usage, joined with && inside ONE gpurun call:
#   tools/gpu.sh test <out> <pytest args...>        pytest -m gpu on a selection, log under gpurun_out/<out>/
#   tools/gpu.sh bench <out> [bench.py args...]     one bench line (no CPU baseline), printed compactly
#   tools/gpu.sh ab <out> "<A> <B> ..." [bench args]   interleaved A/B, each variant twice; a variant is "-" (defaults),
#                                                   "lib:<tag>" (pfotgnrec_amd/lib/libpfotgn_<tag>.so) or "VAR=1,VAR2=x" (environment)
#   tools/gpu.sh stamps <out>                       in-kernel cycle stamps of the layer-1 attention backward (libpfotgn_stamps.so)
#   tools/gpu.sh marks <out> [bench args]           milestone timeline of the step (bench.py --marks)
#   tools/gpu.sh gemm_stamps <out>                  in-kernel stamps of the image / A-stationary / weight-gradient GEMM kernels
#                                                   (build first, here: tools/probes/mkvariant.sh stamps_bxa -DBXA_STAMPS=2)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mode=$1; out=gpurun_out/$2; shift 2; mkdir -p "$out"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d.get('roofline',{}).get('families_ms_per_step',{})
print(d['ms_per_step'], d['config'].get('block_ms_per_step',{}).get('median'), {k:round(v,4) for k,v in f.items()})"; }
case $mode in
  test)
    timeout -k 10 1100 python -m pytest "$@" -m gpu -q > $out/pytest.log 2>&1; rc=$?
    tail -5 $out/pytest.log
    [ $rc -ne 0 ] && grep -n "Error\|assert \|FAILED" $out/pytest.log | head -40
    exit $rc ;;
  bench)
    timeout -k 10 600 python bench.py --no-cpu-baseline "$@" 2>$out/bench.err | tee $out/bench.jsonl | line ;;
  ab)
    variants=$1; shift
    for rep in 1 2; do for v in $variants; do
      echo "== $v"
      envs=""; [ "$v" != "-" ] && envs=${v//,/ }
      case $v in lib:*) envs="PFOTGN_LIB=$GRAFT_REPO_ROOT/pfotgnrec_amd/lib/libpfotgn_${v#lib:}.so" ;; esac
      env $envs timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drop-in --min-seconds 1.0 "$@" 2>/dev/null | line
    done; done 2>&1 | tee $out/ab.txt ;;
  stamps)
    PFOTGN_LIB=$GRAFT_REPO_ROOT/pfotgnrec_amd/lib/libpfotgn_stamps.so timeout -k 10 300 python tools/probes/runs_stamps.py "$@" 2>&1 | grep -v amdgpu.ids | tee $out/stamps.txt ;;
  gemm_stamps)
    export PFOTGN_LIB=$GRAFT_REPO_ROOT/pfotgnrec_amd/lib/libpfotgn_stamps_bxa.so
    { PFO_ASTAT=0 STAMPS2=1 timeout -k 10 200 python tools/probes/areg_stamps.py 53760 704 172 &&
      STAMPS2=1 timeout -k 10 200 python tools/probes/areg_stamps.py 53760 172 704 &&
      ASTAT_SHAPE=1 timeout -k 10 200 python tools/probes/tn_stamps.py &&
      timeout -k 10 200 python tools/probes/tn_stamps.py 704 172 53760; } 2>&1 | grep -v amdgpu.ids | tee $out/gemm_stamps.txt ;;
  marks)
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof --no-drop-in --min-seconds 1.0 --marks 100 "$@" > $out/marks.txt 2>&1; tail -80 $out/marks.txt ;;
  *) echo "unknown mode $mode"; exit 2 ;;
esac
