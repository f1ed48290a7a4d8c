#!/bin/bash
# GPU box: every library given (names under varlociraptor_amd/matrix, without lib/.so) at 4 waves per SIMD against the default build
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/matrix_run.py /tmp/ref.npz quick 2>&1 | tail -1
outs=""
for v in "$@"; do
  w=4; n=$v
  case $v in *@*) w=${v#*@}; n=${v%@*};; esac
  VLR_LIB=$R/varlociraptor_amd/matrix/libvlr_$n.so VLR_WAVES_PER_SIMD=$w python tools/matrix_run.py /tmp/p_${n}_$w.npz quick 2>&1 | tail -1
  outs="$outs /tmp/p_${n}_$w.npz"
done
python tools/o1_probe.py /tmp/ref.npz $outs
for f in $outs; do echo "#### $f"; python tools/o1_probe2.py /tmp/ref.npz $f; done
