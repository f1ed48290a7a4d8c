export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_infl2; N=50000
mkdir -p $O
run() { name=$1; shift; (cd /tmp && rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o p -- python $R/tools/ingest_rate.py $N 32768 device > $O/$name.out 2> $O/$name.err); python $R/tools/pmc_summary.py $(find $O/$name -name "*.db" | head -1) 1 vlr_inflate_kernel > $O/$name.md; find $O/$name -name "*.db" -size +20M -delete; cat $O/$name.md; }
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS
run waits SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
run mem TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum
