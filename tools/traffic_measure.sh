#!/bin/bash
# HBM traffic of the engine kernel per launch (roofline.traffic of bench.py), as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE
# and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel trace only), calibrated on streams of known size in the engine's own
# access widths (4 B/lane f32 column reads, 8 B/lane f64 stores) because the guide's x2 correction is for 16 B/lane reads only.
#   tools/traffic_measure.sh <workload> [loci]      -> profiles/traffic_<workload>.json (stamped with the build id)
export TMPDIR=/tmp
R=$PWD
W=${1:-config3}; shift
LOCI=${1:+--loci $1}
O=$R/gpurun_out/traffic_$W; rm -rf $O; mkdir -p $O
pmc() { name=$1; ctr=$2; shift 2; (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace -d $O/$name -o p -- "$@" > $O/$name.out 2> $O/$name.err); }
pmc cal_read FETCH_SIZE python $R/tools/traffic_calibrate.py read
pmc cal_write WRITE_SIZE python $R/tools/traffic_calibrate.py write
pmc fetch FETCH_SIZE python $R/bench.py --workload $W $LOCI --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-afd
pmc write WRITE_SIZE python $R/bench.py --workload $W $LOCI --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-afd
python $R/tools/traffic_summary.py $O $W
