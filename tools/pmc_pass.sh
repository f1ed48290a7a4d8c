#!/bin/bash
# PMC passes for the engine kernel (each its own rocprofv3 run): instruction mix, utilisation, instruction cache.
#   tools/pmc_pass.sh <outdir under gpurun_out> [workload] [loci]
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1; W=${2:-config3}; N=${3:-50000}
mkdir -p $O
run() { name=$1; shift; (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o p -- python $R/bench.py --workload $W --loci $N --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-afd > $O/$name.json 2> $O/$name.err); python $R/tools/pmc_summary.py $(find $O/$name -name "*.db" | head -1) $N > $O/$name.md; find $O/$name -name "*.db" -size +20M -delete; }
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY
run util SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run f64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
cat $O/insts.md $O/util.md $O/f64.md $O/icache.md
