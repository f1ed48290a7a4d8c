export TMPDIR=/tmp
R=$PWD
cd /tmp && VLR_AFD_LOG_BUDGET_MB=32768 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/afd_diag -o s -- python $R/bench.py --loci 400000 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $R/gpurun_out/afd_diag.json 2> $R/gpurun_out/afd_diag.err
cd $R && python tools/rocpd_summary.py $(find gpurun_out/afd_diag -name "*.db" | head -1) | head -12
python -c "
import json; d=json.load(open('gpurun_out/afd_diag.json')); print(d['value'], d['with_afd'])"
