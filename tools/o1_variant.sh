#!/bin/bash
# One -O1 variant of the call kernel for the bisection of the -O1 / 128-VGPR deviation (DESIGN.md "Build matrix"): only
# vlr_kernels.hip is recompiled with the extra flags, everything else comes from the objects of `make matrix` (build/O1).
#   tools/o1_variant.sh <name> [extra hipcc flags...]   ->  varlociraptor_amd/matrix/libvlr_<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd $R/varlociraptor_amd/csrc
BASE=${BASEFLAGS:--O1 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off}
mkdir -p /tmp/o1v
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE "$@" -c vlr_kernels.hip -o /tmp/o1v/$name.o 2>/dev/null
OBJS=$(ls build/O1/*.o | grep -v "vlr_kernels.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared /tmp/o1v/$name.o $OBJS -o ../matrix/libvlr_$name.so -lz -lpthread -ldl
ls -la ../matrix/libvlr_$name.so
