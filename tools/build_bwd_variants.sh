#!/bin/bash
# Developer builds of librnnt_mi355.so with one part of the fused BACKWARD kernels removed or a knob changed
# (timing ablations; results of ABLATE builds are wrong by design):
#   tools/build_bwd_variants.sh MMA VALU DMA LOAD READ MMA+VALU VALU_PER_MFMA=0  ->  gpurun_tmp/libbwd_<PART>.so
# then e.g.  python tools/with_lib.py gpurun_tmp/libbwd_MMA.so tools/bench_fused.py --only-backward
set -e
cd "$(dirname "$0")/../rnnt_amd/csrc"
make >/dev/null 2>&1
mkdir -p ../../gpurun_tmp
for v in "$@"; do
  defs=$(echo "$v" | tr '+' '\n' | sed -e '/=/s/^/-DFUSED_BWD_/' -e '/=/!{/^STAMP$/!s/^/-DFUSED_BWD_ABLATE_/}' -e 's/^STAMP$/-DFUSED_BWD_STAMP/' | tr '\n' ' ')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $defs -c -o /tmp/fused_bwd_$v.o fused_bwd_mi355.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-soname,librnnt_mi355.so -o ../../gpurun_tmp/libbwd_$v.so ../lib/obj/rnnt_mi355.o ../lib/obj/joint_mi355.o ../lib/obj/fused_mi355.o /tmp/fused_bwd_$v.o
done
