#!/bin/bash
# PMC passes and a kernel trace of the device reader's kernels (inflate, record split, decode) on a synthetic config3 file pair.
#   tools/pmc_inflate.sh <outdir under gpurun_out> [records]
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1; N=${2:-50000}
mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/tools/ingest_rate.py $N 32768 device > $O/trace.out 2> $O/trace.err)
python $R/tools/rocpd_summary.py $(find $O/trace -name "*.db" | head -1) > $O/trace.md 2>&1
run() { name=$1; shift; (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o p -- python $R/tools/ingest_rate.py $N 32768 device > $O/$name.out 2> $O/$name.err); python $R/tools/pmc_summary.py $(find $O/$name -name "*.db" | head -1) 1 vlr_inflate_kernel > $O/$name.md; find $O/$name -name "*.db" -size +20M -delete; }
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY
run util SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
find $O/trace -name "*.db" -size +20M -delete
head -12 $O/trace.md; echo; cat $O/insts.md; echo; cat $O/util.md
