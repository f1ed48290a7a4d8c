#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun): bench lines, kernel trace, PMC passes (each its own run).
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --workload config2 > $O/bench_config2.json 2> $O/bench_config2.err
python tools/rate_all.py 200000 > $O/rate_all.txt 2>&1
python tools/pcie_rate.py 1000000 > $O/pcie_rate.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline > $O/trace_bench.json 2> $O/trace.err)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $O/pmc1 -o p -- python $R/bench.py --loci 50000 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc1.json 2> $O/pmc1.err)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err)
cd $R
for d in trace; do python tools/rocpd_summary.py $(find $O/$d -name "*.db" | head -1) > $O/trace_summary.md; done
python tools/pmc_summary.py $(find $O/pmc1 -name "*.db" | head -1) 50000 > $O/pmc1_summary.md
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*.db" | head -1) > $O/pmc_fetch_summary.md
python tools/pmc_summary.py $(find $O/pmc_write -name "*.db" | head -1) > $O/pmc_write_summary.md
find $O -name "*.db" -size +20M -delete
tail -2 $O/*.md $O/rate_all.txt $O/pcie_rate.txt
tail -c 600 $O/bench_config3.json
