mkdir -p gpurun_out/r04e; O=gpurun_out/r04e
python tools/matrix_run.py /tmp/fast.npz full 2>&1 | tail -1
VLR_NO_FAST_ROOTS=1 python tools/matrix_run.py /tmp/slow.npz full 2>&1 | tail -1
python -c "
import sys; sys.path.insert(0, 'tools')
import matrix_run
print('differing arrays (fast roots on vs off):', matrix_run.compare(['/tmp/slow.npz', '/tmp/fast.npz']))
" > $O/compare.txt 2>&1; tail -5 $O/compare.txt
python tools/rate_variant.py > $O/rate.txt 2>&1
VLR_NO_FAST_ROOTS=1 python tools/rate_variant.py >> $O/rate.txt 2>&1; cat $O/rate.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_golden_synth.py tests/test_gpu_properties.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
