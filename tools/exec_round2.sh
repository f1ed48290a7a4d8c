#!/bin/bash
# Second diagnosis run of round 6: (1) is the max-ILP build with fresh_lane bit-identical to the shipped one once fresh_sd's outputs are
# early-clobber?  (2) where does the -O1 build part at four waves per SIMD (trace with the coefficient terms)?
R=$PWD; M=$R/varlociraptor_amd/matrix; O=$R/gpurun_out/r06_exec2
mkdir -p $O
run() { VLR_LIB=$1 VLR_WAVES_PER_SIMD=$3 timeout 600 python tools/matrix_run.py $2 quick > /dev/null 2>&1 || echo "matrix_run FAILED for $1"; }
{
echo "== shipped build vs max-ILP with fresh_lane"
run $R/varlociraptor_amd/libvlr.so $O/def.npz
run $M/libvlr_ilpf.so $O/ilpf.npz
python tools/exec_trace_run.py first $O/def.npz $O/ilpf.npz
for w in 2 4; do run $M/libvlr_ilpf.so $O/ilpf_$w.npz $w; python tools/exec_trace_run.py first $O/def.npz $O/ilpf_$w.npz | tail -1; done
echo "== -O1 at four waves per SIMD (no instrumentation)"
run $R/varlociraptor_amd/libvlr.so $O/def4.npz 4
run $M/libvlr_O1plain.so $O/O1_4.npz 4
python tools/exec_trace_run.py first $O/def.npz $O/def4.npz | tail -1
python tools/exec_trace_run.py first $O/def4.npz $O/O1_4.npz
echo "== -O1 at four waves per SIMD, trace builds"
run $M/libvlr_tr_def.so $O/tr_def4.npz 4
run $M/libvlr_tr_O1.so $O/tr_O1_4.npz 4
python tools/exec_trace_run.py first $O/tr_def4.npz $O/tr_O1_4.npz | tee $O/first_O1.txt
set -- $(grep '^FIRST' $O/first_O1.txt)
if [ "$2" != "None" ]; then
  VLR_WAVES_PER_SIMD=4 VLR_LIB=$M/libvlr_tr_def.so timeout 300 python tools/exec_trace_run.py trace $2 $3 $O/trace_def4.npz
  VLR_WAVES_PER_SIMD=4 VLR_LIB=$M/libvlr_tr_O1.so timeout 300 python tools/exec_trace_run.py trace $2 $3 $O/trace_O1_4.npz
  python tools/exec_trace_run.py diff $O/trace_def4.npz $O/trace_O1_4.npz 24
fi
} 2>&1 | tee $O/log.txt
{
echo "== kernel time, 200 000 loci: shipped schedule vs max-ILP (both with fresh_lane)"
timeout 600 python tools/rate_variant.py
VLR_LIB=$M/libvlr_ilpf.so timeout 600 python tools/rate_variant.py
} 2>&1 | tee -a $O/log.txt
