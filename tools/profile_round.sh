#!/bin/bash
# Everything the round's profile note is written from, in one GPU-box call:
#   tools/profile_round.sh <tag>     -> gpurun_out/<tag>/{bench_*.json, stats_*, pmc/, traffic}
# (bench lines of every workload, rocprofv3 --kernel-trace --stats of the default bench command, the PMC passes of
# tools/pmc_pass.sh and the calibrated HBM traffic of tools/traffic_measure.sh.)
export TMPDIR=/tmp
R=$PWD
T=${1:-round}
O=$R/gpurun_out/$T; mkdir -p $O
python -c "from varlociraptor_amd import engine; print(engine.build_id())" > $O/build_id.txt
for W in config3 config2 config4 config5 realign; do
  python bench.py --workload $W > $O/bench_$W.json 2> $O/bench_$W.err
done
python bench.py --workload config3 --afd --no-cpu-baseline > $O/bench_config3_afd.json 2> $O/bench_config3_afd.err
python bench.py --workload cli --steps 3 --warmup 1 > $O/bench_cli.json 2> $O/bench_cli.err
python bench.py --workload cli --loci 1000000 --steps 2 --warmup 1 > $O/bench_cli_1M.json 2> $O/bench_cli_1M.err
VLR_INGEST_HOST=1 python bench.py --workload cli --steps 3 --warmup 1 > $O/bench_cli_hostreader.json 2> $O/bench_cli_hostreader.err
python bench.py --workload ingest --steps 3 --warmup 1 > $O/bench_ingest.json 2> $O/bench_ingest.err
python tools/cpu_probe.py > $O/cpu_probe.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/stats_config3 -o s -- python $R/bench.py --no-cpu-baseline --no-end-to-end --no-afd > $O/stats_config3.json 2> $O/stats_config3.err)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/stats_config3_afd -o s -- python $R/bench.py --afd --no-cpu-baseline --no-end-to-end > $O/stats_config3_afd.json 2> $O/stats_config3_afd.err)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/stats_realign -o s -- python $R/bench.py --workload realign --no-cpu-baseline > $O/stats_realign.json 2> $O/stats_realign.err)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/stats_cli -o s -- python $R/bench.py --workload cli --steps 3 --warmup 1 > $O/stats_cli.json 2> $O/stats_cli.err)
DBC=$(find $O/stats_cli -name "*.db" | head -1)
if [ -n "$DBC" ]; then python tools/rocpd_summary.py $DBC | head -20 > $O/stats_cli.md; echo >> $O/stats_cli.md; python tools/timeline_busy.py $DBC 0.4 1.0 >> $O/stats_cli.md; fi
find $O -name "*.db" -size +20M -delete
bash tools/pmc_pass.sh $T/pmc config3 50000 > $O/pmc.md 2>&1
bash tools/pmc_inflate.sh $T/pmc_inflate 50000 > $O/pmc_inflate.md 2>&1
bash tools/traffic_measure.sh config3 > $O/traffic.out 2>&1
cp $R/profiles/traffic_config3.json $O/ 2>/dev/null
ls $O
