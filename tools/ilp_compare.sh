R=$PWD; M=$R/varlociraptor_amd/matrix
timeout 200 python tools/matrix_run.py /tmp/def.npz quick > /dev/null 2>&1
for v in "$@"; do
  if [ "$v" = "nofast" ]; then VLR_NO_FAST_ROOTS=1 VLR_LIB=$M/libvlr_ilp.so timeout 200 python tools/matrix_run.py /tmp/$v.npz quick > /dev/null 2>&1
  else VLR_LIB=$M/libvlr_$v.so timeout 200 python tools/matrix_run.py /tmp/$v.npz quick > /dev/null 2>&1; fi
  python -c "
import sys; sys.path.insert(0, 'tools')
import matrix_run, io, contextlib
f = io.StringIO()
with contextlib.redirect_stdout(f):
    n = matrix_run.compare(['/tmp/def.npz', '/tmp/$v.npz'])
print('$v', 'differing arrays:', n)
"
done
