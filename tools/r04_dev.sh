mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_ingest_device.py -x -q > gpurun_out/r04d/pytest_dev.txt 2>&1; tail -3 gpurun_out/r04d/pytest_dev.txt
timeout 600 python tools/ingest_rate.py 100000 32768 device 2>&1 | tail -2
timeout 900 python bench.py --workload cli > gpurun_out/r04d/bench_cli.json 2> gpurun_out/r04d/bench_cli.err; python -c "
import json; d=json.load(open('gpurun_out/r04d/bench_cli.json')); print(d['value'], d['stages_s'], d['native_stage_seconds_per_step'])"
tail -3 gpurun_out/r04d/bench_cli.err
bash tools/pmc_inflate.sh r04d_pmc 50000 2>&1 | grep -E "inflate_kernel|decode_kernel|SQ_INSTS_SALU|SQ_INSTS_VALU|SQ_WAVE_CYCLES|SQ_WAVES|SQ_INSTS_LDS"
