mkdir -p gpurun_out/r04f; O=gpurun_out/r04f
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['with_afd']['value'], d['pcie_inclusive'], d['end_to_end'])"
python bench.py --workload cli --steps 3 --warmup 1 > $O/bench_cli.json 2> $O/bench_cli.err; python -c "
import json; d=json.load(open('$O/bench_cli.json')); print(d['value'], d['stages_s'], d['native_stage_seconds_per_step'])"
python -m pytest tests/test_gpu_cli_end_to_end.py tests/test_ingest.py -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
