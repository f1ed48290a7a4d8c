# one GPU-box call for the round's profile note: the GPU suite + tools/profile_round.sh (bench lines of every workload incl. the front door,
# kernel traces, PMC passes of the call kernel and of the inflate kernel, calibrated HBM traffic)
T=${1:-r05a}
mkdir -p gpurun_out/$T
python tools/rate_variant.py 2>&1 | grep config | tee gpurun_out/$T/rates.txt
python -m pytest tests -m gpu -q > gpurun_out/$T/pytest.txt 2>&1; tail -3 gpurun_out/$T/pytest.txt
bash tools/profile_round.sh $T > gpurun_out/$T/profile_round.log 2>&1
python bench.py --workload realign --mode homopolymer > gpurun_out/$T/bench_realign_homopolymer.json 2> /dev/null
python bench.py --workload realign --mode fast > gpurun_out/$T/bench_realign_fast.json 2> /dev/null
ls gpurun_out/$T | head -60
