#!/bin/bash
# Build one named engine variant into varlociraptor_amd/matrix/libvlr_<name>.so
#   tools/build_variant.sh <name> [extra hipcc flags...]        e.g.  tools/build_variant.sh sync -DVLR_WB_SYNC
# BASEFLAGS may be overridden from the environment (default = the Makefile's CXXFLAGS).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
BASE=${BASEFLAGS:--O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -disable-machine-licm}
mkdir -p $R/varlociraptor_amd/matrix
cd $R/varlociraptor_amd/csrc
SRC=$(grep '^SRC = ' Makefile | cut -d= -f2)
LIBS=$(grep '^LIBS = ' Makefile | cut -d= -f2)
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BASE "$@" -DVLR_SRC_ID="\"$(cat $SRC vlr_plan.h vlr_gpuio.h ../../include/vlr.h ../../include/vlr_detmath.h | sha1sum | cut -c1-16)\"" -shared $SRC -o ../matrix/libvlr_$name.so $LIBS 2>&1 | grep -v "warning\|unused\|\^\|^ *[0-9]* |\|generated" || true
ls -la ../matrix/libvlr_$name.so
