#!/bin/bash
# PMC pass: what the waves wait for (in-flight levels per memory type, instruction fetch, branches).  tools/pmc_wait.sh <outdir> [workload] [loci]
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1; W=${2:-config3}; N=${3:-50000}
mkdir -p $O
run() { name=$1; shift; (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o p -- python $R/bench.py --workload $W --loci $N --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-afd > $O/$name.json 2> $O/$name.err); python $R/tools/pmc_summary.py $(find $O/$name -name "*.db" | head -1) $N > $O/$name.md; find $O/$name -name "*.db" -size +20M -delete; }
run level SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM
run misc SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY
cat $O/level.md $O/misc.md
