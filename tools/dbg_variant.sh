#!/bin/bash
# One diagnosis variant of the call kernel (round 6, VERDICT r05 "next" #1): only vlr_kernels.hip is recompiled with the extra flags,
# everything else comes from the objects of `make` (build/default).  STRAT=ilp adds the machine scheduler's max-ILP strategy (the
# configuration that returns other results with fresh_lane), STRAT=O1W4 builds -O1 (compare at VLR_WAVES_PER_SIMD=4).
#   STRAT=default|ilp|O1 tools/dbg_variant.sh <name> [extra hipcc flags...]   ->  varlociraptor_amd/matrix/libvlr_<name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd $R/varlociraptor_amd/csrc
BASE="-O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -disable-machine-licm"
case "${STRAT:-default}" in
  ilp) BASE="$BASE -mllvm -amdgpu-sched-strategy=max-ilp" ;;
  O1) BASE="-O1 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off" ;;
esac
mkdir -p /tmp/dbgv ../matrix
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE "$@" -c vlr_kernels.hip -o /tmp/dbgv/$name.o 2>/dev/null
OBJS=$(ls build/default/*.o | grep -v "vlr_kernels.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared /tmp/dbgv/$name.o $OBJS -o ../matrix/libvlr_$name.so -lz -lpthread -ldl
ls -la ../matrix/libvlr_$name.so
