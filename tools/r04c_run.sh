mkdir -p gpurun_out/r04c; O=gpurun_out/r04c
VLR_LIB=$PWD/varlociraptor_amd/matrix/libvlr_prev.so python tools/rate_variant.py > $O/rate.txt 2>&1
python tools/rate_variant.py >> $O/rate.txt 2>&1; cat $O/rate.txt
VLR_LIB=$PWD/varlociraptor_amd/matrix/libvlr_prof.so python tools/profile_phases.py config3 50000 > $O/cycles_config3.txt 2>&1; tail -3 $O/cycles_config3.txt
python -m pytest tests/test_gpu_cli_end_to_end.py tests/test_gpu_node.py tests/test_ingest.py tests/test_fdr.py -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --workload cli --steps 3 --warmup 1 > $O/bench_cli.json 2> $O/bench_cli.err; python -c "
import json; d=json.load(open('$O/bench_cli.json')); print(d['value'], d['stages_s'], d['native_stage_seconds_per_step'])"
