mkdir -p gpurun_out/r04d; O=gpurun_out/r04d
python -m pytest tests -m gpu -q --durations=12 > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt
python tools/rate_variant.py > $O/rate.txt 2>&1; cat $O/rate.txt
