#!/bin/bash
# Build varlociraptor_amd/matrix/libvlr_valuprof.so: the engine with per-region VALU instruction counters (see PROF_ADD under
# VLR_PROFILE_VALU in vlr_kernels.hip and tools/valu_instrument.py).  Run: VLR_LIB=.../libvlr_valuprof.so python tools/profile_phases.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
L=/opt/rocm/lib/llvm/bin
# usage: tools/valu_profile.sh [class]   (class: see VALU_CLASS in tools/valu_instrument.py; the library is libvlr_valuprof[_class].so)
CLS=${1:-all}
SUF=$([ $CLS = all ] && echo "" || echo "_$CLS")
W=/tmp/valuprof$SUF; rm -rf $W; mkdir -p $W $R/varlociraptor_amd/matrix
cd $R/varlociraptor_amd/csrc
F="-O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -disable-machine-licm -DVLR_PROFILE -DVLR_PROFILE_VALU"
SRC=$(grep '^SRC = ' Makefile | cut -d= -f2)
ID="-DVLR_SRC_ID=\"$(cat $SRC vlr_plan.h vlr_gpuio.h ../../include/vlr.h ../../include/vlr_detmath.h | sha1sum | cut -c1-16)\""
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F "$ID" -S --cuda-device-only vlr_kernels.hip -o $W/dev.s 2>/dev/null
VALU_CLASS=$CLS python $R/tools/valu_instrument.py $W/dev.s $W/dev_i.s
python $R/tools/asm_islands.py $W/dev_i.s $W/dev.gpuo
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.co $W/dev.gpuo
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.co -output=$W/dev.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F "$ID" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c vlr_kernels.hip -o $W/k.o 2>/dev/null
for s in $SRC; do
  [ $s = vlr_kernels.hip ] && continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $F "$ID" -c $s -o $W/$(basename $s).o 2>/dev/null
done
LIBS=$(grep '^LIBS = ' Makefile | cut -d= -f2)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $W/*.o -o ../matrix/libvlr_valuprof$SUF.so $LIBS 2>/dev/null
ls -la ../matrix/libvlr_valuprof$SUF.so
