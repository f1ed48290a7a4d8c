# one GPU-box call for the round's profile note: tools/profile_round.sh + per-region VALU / cycle profiles + the GPU suite
T=${1:-r04a}
mkdir -p gpurun_out/$T
[ -f varlociraptor_amd/matrix/libvlr_prev.so ] && VLR_LIB=$PWD/varlociraptor_amd/matrix/libvlr_prev.so python tools/rate_variant.py 2>&1 | grep config3 | sed 's/^/prev: /'
python tools/rate_variant.py 2>&1 | grep config | tee gpurun_out/$T/rates.txt
python -m pytest tests -m gpu -q > gpurun_out/$T/pytest.txt 2>&1; tail -3 gpurun_out/$T/pytest.txt
bash tools/profile_round.sh $T > gpurun_out/$T/profile_round.log 2>&1
VLR_LIB=$PWD/varlociraptor_amd/matrix/libvlr_valuprof.so python tools/profile_phases.py config3 50000 > gpurun_out/$T/valu_by_region.txt 2>&1
VLR_LIB=$PWD/varlociraptor_amd/matrix/libvlr_prof.so python tools/profile_phases.py config3 50000 > gpurun_out/$T/cycles_by_region.txt 2>&1
python bench.py --workload realign --mode homopolymer > gpurun_out/$T/bench_realign_homopolymer.json 2> /dev/null
python bench.py --workload realign --mode fast > gpurun_out/$T/bench_realign_fast.json 2> /dev/null
ls gpurun_out/$T | head -40
