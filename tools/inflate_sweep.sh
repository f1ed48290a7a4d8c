#!/bin/bash
# Table-size / ring-size sweep of the inflate kernel (round 6): LDS per wave decides how many members a CU holds in flight, and the
# decoder is bound by the latencies of ONE wave.  One library per variant (only vlr_inflate.hip recompiled, the other objects of the
# default build linked in): varlociraptor_amd/matrix/libvlr_i_<name>.so; `tools/inflate_sweep.sh run` times them on the GPU box.
cd "$(dirname "$0")/.."
if [ "$1" = run ]; then
  for lib in default $(ls varlociraptor_amd/matrix/ | grep '^libvlr_i_'); do
    if [ $lib = default ]; then unset VLR_LIB; else export VLR_LIB=$PWD/varlociraptor_amd/matrix/$lib; fi
    echo "== $lib"; python tools/ingest_rate.py ${2:-200000} 65536 device 2>&1 | grep "device reader" | tail -2
  done
  exit 0
fi
C=varlociraptor_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -mllvm -disable-machine-licm"
build() { name=$1; shift
  mkdir -p $C/build/i_$name
  src=$C/vlr_inflate.hip
  if [ -n "$FLUSH" ]; then   # (the flush piece is not a -D knob of the shipped source: a patched copy beside it)
    src=$C/build/i_$name/vlr_inflate_flush.hip; sed "s/kFlush = 1024;/kFlush = $FLUSH;/" $C/vlr_inflate.hip > $src
  fi
  /opt/rocm/bin/hipcc $FLAGS "$@" -I$C -c $src -o $C/build/i_$name/vlr_inflate.hip.o || return 1
  objs=$(ls $C/build/default/*.o | grep -v vlr_inflate)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $objs $C/build/i_$name/vlr_inflate.hip.o -o varlociraptor_amd/matrix/libvlr_i_$name.so -lz -lpthread -ldl
}
# first sweep (profiles/r06g_experiments.md section 2): l9d8, l9d9, l8d8, l11d9 (-DVLR_INFL_LIT_BITS / -DVLR_INFL_DIST_BITS), l9d8r8 (+ -DVLR_INFL_RING=8192)
FLUSH=512 build r2f5 -DVLR_INFL_RING=2048 &
FLUSH=512 build r4f5 &
FLUSH=512 build r2f5l9 -DVLR_INFL_RING=2048 -DVLR_INFL_LIT_BITS=9 &
wait
ls -la varlociraptor_amd/matrix/ | grep libvlr_i_
