#!/bin/bash
# Diagnosis run on the GPU box: every engine variant under varlociraptor_amd/matrix x register budgets, bitwise vs base.
R=$PWD
O=$R/gpurun_out/matrix
rm -rf $O; mkdir -p $O
MODE=${MODE:-quick}
python tools/matrix_run.py $O/base.npz $MODE > $O/log.txt 2>&1
for lib in ${VARIANTS:-fence plain sync nouni nok4 O1 licm nocse}; do
  for w in ${BUDGETS:-2 3 4}; do
    VLR_LIB=$R/varlociraptor_amd/matrix/libvlr_$lib.so VLR_WAVES_PER_SIMD=$w timeout 600 python tools/matrix_run.py $O/${lib}_$w.npz $MODE >> $O/log.txt 2>&1 || echo "FAILED $lib $w" >> $O/log.txt
  done
done
python tools/matrix_run.py compare $O/base.npz $O/*_*.npz > $O/compare.txt 2>&1
cat $O/compare.txt | grep -v "^     " 
