# one GPU-box call for the round's profile note (round 6): kernel rates, the GPU suite, tools/profile_round.sh (bench lines of every
# workload incl. the front door, kernel traces, PMC passes, calibrated HBM traffic) and the VALU stamp of the call kernel
# (tools/valu_stamp.py -> gpurun_out/valu_config3.json: valu_busy / f64_share of bench.py's roofline object)
T=${1:-r06h}
mkdir -p gpurun_out/$T
python tools/rate_variant.py 2>&1 | grep config | tee gpurun_out/$T/rates.txt
python -m pytest tests -m gpu -q > gpurun_out/$T/pytest.txt 2>&1; tail -3 gpurun_out/$T/pytest.txt
bash tools/profile_round.sh $T > gpurun_out/$T/profile_round.log 2>&1
python tools/valu_stamp.py gpurun_out/$T/pmc config3 50000 > gpurun_out/$T/valu_stamp.log 2>&1
cp gpurun_out/valu_config3.json gpurun_out/traffic_config3.json gpurun_out/$T/ 2>/dev/null
python bench.py --workload realign --mode homopolymer > gpurun_out/$T/bench_realign_homopolymer.json 2> /dev/null
python bench.py --workload realign --mode fast > gpurun_out/$T/bench_realign_fast.json 2> /dev/null
# the default line once more, now that the stamps of this build exist (as the driver will run it)
cp gpurun_out/valu_config3.json gpurun_out/traffic_config3.json profiles/ 2>/dev/null
python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err
ls gpurun_out/$T | head -80
