#!/bin/bash
# One variant of the call kernel under -mllvm -amdgpu-sched-strategy=max-ilp (round 5: that scheduler strategy makes the call kernel return
# other results while every other kernel of the library passes its tests — DESIGN.md "Build matrix"): only vlr_kernels.hip is recompiled
# with the extra flags, everything else comes from the objects of `make` (build/default).
#   tools/ilp_variant.sh <name> [extra hipcc flags...]   ->  varlociraptor_amd/matrix/libvlr_<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd $R/varlociraptor_amd/csrc
BASE=${BASEFLAGS:--O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -disable-machine-licm}
STRAT=${STRAT:--mllvm -amdgpu-sched-strategy=max-ilp}
mkdir -p /tmp/ilpv
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE $STRAT "$@" -c vlr_kernels.hip -o /tmp/ilpv/$name.o 2>/dev/null
OBJS=$(ls build/default/*.o | grep -v "vlr_kernels.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared /tmp/ilpv/$name.o $OBJS -o ../matrix/libvlr_$name.so -lz -lpthread -ldl
ls -la ../matrix/libvlr_$name.so
