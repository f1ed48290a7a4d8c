#!/bin/bash
# GPU-box run of the round-6 diagnosis builds (VERDICT r05 "next" #1): (a) the EXEC / uniformity assert build under the default and the
# max-ILP strategy, (b) the per-wave traces of the first locus whose results differ between two builds of the same source.
# Libraries: tools/dbg_variant.sh (built in the container, they travel with the snapshot).  Output: gpurun_out/r06_exec/.
R=$PWD; M=$R/varlociraptor_amd/matrix; O=$R/gpurun_out/r06_exec
mkdir -p $O
run() { VLR_LIB=$1 VLR_WAVES_PER_SIMD=$3 timeout 600 python tools/matrix_run.py $2 quick > /dev/null 2>&1 || echo "matrix_run FAILED for $1"; }
{
echo "== shipped build vs max-ILP with fresh_lane (no instrumentation)"
run $R/varlociraptor_amd/libvlr.so $O/def.npz
run $M/libvlr_ilpf.so $O/ilpf.npz
python tools/exec_trace_run.py first $O/def.npz $O/ilpf.npz
echo "== EXEC assert builds"
VLR_LIB=$M/libvlr_xa_def.so timeout 900 python tools/exec_trace_run.py assert $O/xa_def.json quick
VLR_LIB=$M/libvlr_xa_ilp.so timeout 900 python tools/exec_trace_run.py assert $O/xa_ilp.json quick
echo "== trace builds: do they still part?"
run $M/libvlr_tr_def.so $O/tr_def.npz
run $M/libvlr_tr_ilp.so $O/tr_ilp.npz
python tools/exec_trace_run.py first $O/def.npz $O/tr_def.npz | tail -1
python tools/exec_trace_run.py first $O/tr_def.npz $O/tr_ilp.npz | tee $O/first_ilp.txt
set -- $(grep '^FIRST' $O/first_ilp.txt)
if [ "$2" != "None" ]; then
  VLR_LIB=$M/libvlr_tr_def.so timeout 300 python tools/exec_trace_run.py trace $2 $3 $O/trace_def.npz
  VLR_LIB=$M/libvlr_tr_ilp.so timeout 300 python tools/exec_trace_run.py trace $2 $3 $O/trace_ilp.npz
  python tools/exec_trace_run.py diff $O/trace_def.npz $O/trace_ilp.npz 16
fi
echo "== -O1 at four waves per SIMD"
run $M/libvlr_tr_def.so $O/tr_def4.npz 4
run $M/libvlr_tr_O1.so $O/tr_O1_4.npz 4
python tools/exec_trace_run.py first $O/tr_def4.npz $O/tr_O1_4.npz | tee $O/first_O1.txt
set -- $(grep '^FIRST' $O/first_O1.txt)
if [ "$2" != "None" ]; then
  VLR_WAVES_PER_SIMD=4 VLR_LIB=$M/libvlr_tr_def.so timeout 300 python tools/exec_trace_run.py trace $2 $3 $O/trace_def4.npz
  VLR_WAVES_PER_SIMD=4 VLR_LIB=$M/libvlr_tr_O1.so timeout 300 python tools/exec_trace_run.py trace $2 $3 $O/trace_O1_4.npz
  python tools/exec_trace_run.py diff $O/trace_def4.npz $O/trace_O1_4.npz 16
fi
} 2>&1 | tee $O/log.txt
