#!/bin/bash
# A/B library with ONE source rebuilt (default gemm.hip; SRC=attn tools/probes/mkvariant.sh ... for another):
#   tools/probes/mkvariant.sh <tag> [-DNAME=value ...] -> pfotgnrec_amd/lib/libpfotgn_<tag>.so
# (the other objects come from the default build: run python -m pfotgnrec_amd.build first; use with PFOTGN_LIB=... or tools/gpu.sh ab "lib:<tag>")
# e.g. -DBXA_ABL=<bits> (timing-only ablations), -DBXA_STAMPS=2 (in-kernel stamps: tools/gpu.sh gemm_stamps), -DBXA_EPI=0 (round-3 epilogues)
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(git -C "$(dirname "$0")" rev-parse --show-toplevel)}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRC=${SRC:-gemm}
L=$ROOT/pfotgnrec_amd/lib
mkdir -p $L/obj_$tag
$HIPCC -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -c $ROOT/pfotgnrec_amd/csrc/$SRC.hip -o $L/obj_$tag/$SRC.o || exit 1
objs=""
for f in sampler gemm attn memory misc csr tgn; do
  if [ "$f" = "$SRC" ]; then objs="$objs $L/obj_$tag/$SRC.o"; else objs="$objs $L/obj/$f.o"; fi
done
$HIPCC -shared -fPIC --offload-arch=gfx950 -o $L/libpfotgn_$tag.so $objs
echo built $tag
