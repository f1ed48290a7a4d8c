mkdir -p gpurun_out/r04g
timeout 1200 python -m pytest tests/test_gpu_ingest_device.py tests/test_gpu_cli_end_to_end.py -x -q > gpurun_out/r04g/pytest_dev.txt 2>&1; tail -3 gpurun_out/r04g/pytest_dev.txt
for n in 200000 1000000; do timeout 1500 python bench.py --workload cli --loci $n > gpurun_out/r04g/bench_cli_n$n.json 2> gpurun_out/r04g/bench_cli.err; python -c "
import json; d=json.load(open('gpurun_out/r04g/bench_cli_n$n.json')); print($n, d['value'], d['ms_per_step'], d['stages_s'])"; done
