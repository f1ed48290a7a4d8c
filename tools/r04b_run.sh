mkdir -p gpurun_out/r04b; O=gpurun_out/r04b
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
bash tools/compare_prev.sh full > $O/compare.txt 2>&1; tail -22 $O/compare.txt
VLR_LIB=$PWD/varlociraptor_amd/matrix/libvlr_valuprof.so python tools/profile_phases.py config3 50000 > $O/valu_config3.txt 2>&1
VLR_LIB=$PWD/varlociraptor_amd/matrix/libvlr_valuprof.so python tools/profile_phases.py config2 50000 > $O/valu_config2.txt 2>&1
python bench.py --workload realign --mode homopolymer > $O/bench_realign_homopolymer.json 2> $O/bench_realign_homopolymer.err; tail -c 600 $O/bench_realign_homopolymer.json
