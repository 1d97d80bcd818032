#!/bin/bash
# A/B library with only gemm.hip rebuilt: tools/probes/mkvariant.sh <tag> [-DNAME=value ...] -> pfotgnrec_amd/lib/libpfotgn_<tag>.so
# (the other objects come from the default build: run python -m pfotgnrec_amd.build first; use with PFOTGN_LIB=... or tools/gpu.sh ab "lib:<tag>")
# e.g. -DBXA_ABL=<bits> (timing-only ablations), -DBXA_STAMPS=2 (in-kernel stamps: tools/gpu.sh gemm_stamps), -DBXA_EPI=0 (round-3 epilogues)
tag=$1; shift
L=/root/repo/pfotgnrec_amd/lib
mkdir -p $L/obj_$tag
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 "$@" -c /root/repo/pfotgnrec_amd/csrc/gemm.hip -o $L/obj_$tag/gemm.o || exit 1
objs=""
for f in sampler attn memory misc csr tgn; do objs="$objs $L/obj/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libpfotgn_$tag.so $objs $L/obj_$tag/gemm.o
echo built $tag
