#!/bin/bash
# Bit-for-bit comparison of the working tree's engine against varlociraptor_amd/matrix/libvlr_prev.so (the previous commit's
# kernel, built by hand from a git worktree with tools/build_variant.sh prev) on the build-matrix workloads, and the kernel
# times of both.   usage (GPU box): bash tools/compare_prev.sh [quick|full]
mode=${1:-full}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/matrix_run.py /tmp/cur.npz $mode 2>&1 | tail -1
VLR_LIB=$R/varlociraptor_amd/matrix/libvlr_prev.so python tools/matrix_run.py /tmp/prev.npz $mode 2>&1 | tail -1
python -c "
import sys; sys.path.insert(0, 'tools')
import matrix_run
print('differing arrays:', matrix_run.compare(['/tmp/prev.npz', '/tmp/cur.npz'], ignore_build_id=True))
print('beyond 1e-9:', matrix_run.compare_tol(['/tmp/prev.npz', '/tmp/cur.npz']))
"
VLR_LIB=$R/varlociraptor_amd/matrix/libvlr_prev.so python tools/rate_variant.py 2>/dev/null
python tools/rate_variant.py 2>/dev/null
