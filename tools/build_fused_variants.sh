#!/bin/bash
# Developer builds of librnnt_mi355.so with one part of the fused projection kernel removed (timing ablations):
#   tools/build_fused_variants.sh DMA STORE EXP MMA MMA+STORE  ->  gpurun_tmp/lib_no_<PART>.so   (results are wrong by design)
set -e
cd "$(dirname "$0")/../rnnt_amd/csrc"
make >/dev/null 2>&1
mkdir -p ../../gpurun_tmp
for v in "$@"; do
  # MMA+STORE -> -DFUSED_ABLATE_MMA -DFUSED_ABLATE_STORE; a part with '=' is a plain define: PF=4 -> -DFUSED_PF=4
  defs=$(echo "$v" | tr '+' '\n' | sed -e '/=/s/^/-DFUSED_/' -e '/=/!s/^/-DFUSED_ABLATE_/' | tr '\n' ' ')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -c -o /tmp/fused_$v.o fused_mi355.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-soname,librnnt_mi355.so -o ../../gpurun_tmp/lib_no_$v.so ../lib/obj/rnnt_mi355.o ../lib/obj/joint_mi355.o ../lib/obj/fused_bwd_mi355.o /tmp/fused_$v.o
done
