"""Bisection aid for the -O1 / 128-VGPR deviation (DESIGN.md "Build matrix"): compare dumps of tools/matrix_run.py and describe HOW a
deviating build deviates (which arrays, relative size, whether the deviation of ln_marginal is proportional to the value).
usage: python tools/o1_probe.py ref.npz other.npz [other.npz ...]"""
import sys
import numpy as np
ref = np.load(sys.argv[1])
for path in sys.argv[2:]:
    o = np.load(path)
    nd = 0; worst = (0.0, ""); rel_all = []
    for k in ref.files:
        if k == "build_id" or k not in o.files: continue
        a, b = ref[k], o[k]
        if a.shape != b.shape: nd += 1; continue
        same = np.array_equal(a.view(np.uint8) if a.dtype.kind == "f" else a, b.view(np.uint8) if b.dtype.kind == "f" else b)
        if same: continue
        nd += 1
        if a.dtype.kind == "f" and k.endswith("/ln_marginal"):
            fin = np.isfinite(a) & np.isfinite(b) & (a != 0)
            if fin.any():
                r = (b[fin] - a[fin]) / np.abs(a[fin])
                rel_all.append(r)
                if np.abs(r).max() > worst[0]: worst = (float(np.abs(r).max()), k)
    msg = "%-28s differing arrays %3d" % (path.split("/")[-1], nd)
    if rel_all:
        r = np.concatenate(rel_all)
        msg += "  ln_marginal rel dev: median %.3g  p10 %.3g  p90 %.3g  max|.| %.3g (%s)  frac nonzero %.2f" % (
            np.median(r), np.percentile(r, 10), np.percentile(r, 90), worst[0], worst[1], float((r != 0).mean()))
    print(msg, flush=True)
