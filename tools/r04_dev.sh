mkdir -p gpurun_out/r04g
timeout 600 python bench.py --workload ingest --no-cpu-baseline > gpurun_out/r04g/bench_ingest.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04g/bench_ingest.json')); print('ingest', d['value'], d['stages_s'])"
for n in 200000 1000000; do timeout 1500 python bench.py --workload cli --loci $n > gpurun_out/r04g/bench_cli_n$n.json 2> gpurun_out/r04g/bench_cli.err; python -c "
import json; d=json.load(open('gpurun_out/r04g/bench_cli_n$n.json')); print($n, d['value'], d['ms_per_step'], d['stages_s'])"; done
