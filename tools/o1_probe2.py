"""Which loci and which events deviate between two dumps of tools/matrix_run.py (bisection of the -O1 / 128-VGPR deviation).
usage: python tools/o1_probe2.py ref.npz other.npz [workload ...]"""
import sys
import numpy as np
ref, o = np.load(sys.argv[1]), np.load(sys.argv[2])
wl = sys.argv[3:] or ["config2", "config3", "single_3", "config5", "tn_tiny"]
for w in wl:
    a_m, b_m = ref[w + "/ln_marginal"], o[w + "/ln_marginal"]
    a_p, b_p = ref[w + "/ln_posterior"], o[w + "/ln_posterior"]
    st = ref[w + "/status"]
    ja, jb = a_p + a_m[:, None], b_p + b_m[:, None]          # event values (joint), last column = artifact posterior + marginal (ignore)
    with np.errstate(invalid="ignore"):
        d = jb - ja
    d = np.where(np.isfinite(d), d, 0.0)
    dm = b_m - a_m
    big = np.argsort(-np.abs(dm))[:6]
    print("== %s: n=%d  loci with |d ln_marginal| > 1e-9: %d, > 1e-12: %d; max %.3g" % (w, len(a_m), int((np.abs(dm) > 1e-9).sum()), int((np.abs(dm) > 1e-12).sum()), float(np.abs(dm).max())))
    for i in big:
        print("   locus %4d status %3d  marginal %.6f  d_marginal %+.3e  d_event_values %s" % (i, st[i], a_m[i], dm[i], " ".join("%+.2e" % v for v in d[i, :-1])))
    # relation to the value: d / value and d / (number of doublings)
    nz = np.abs(dm) > 1e-9
    if nz.any():
        print("   d/value over the deviating loci: min %.3g median %.3g max %.3g" % tuple(np.percentile(dm[nz] / a_m[nz], [0, 50, 100])))
# the visited points of the MAP chain as the AFD lists carry them: x positions and joint values (ln prob + marginal) per point
for w in wl:
    if w + "/afd_vaf" not in ref.files:
        continue
    xa, xb = ref[w + "/afd_vaf"], o[w + "/afd_vaf"]
    la, lb = ref[w + "/afd_lnprob"] + ref[w + "/ln_marginal"][:, None, None], o[w + "/afd_lnprob"] + o[w + "/ln_marginal"][:, None, None]
    ca, cb = ref[w + "/afd_count"], o[w + "/afd_count"]
    same_cnt = np.array_equal(ca, cb)
    dx = np.abs(xa - xb).max() if xa.shape == xb.shape else -1
    with np.errstate(invalid="ignore"):
        dj = np.abs(la - lb)
    dj = np.where(np.isfinite(dj), dj, 0.0)
    print("== %s AFD lists: counts equal %s, max |dx| %.3g, max |d joint of a visited point| %.3g, points off by > 1e-9: %d of %d" % (
        w, same_cnt, dx, float(dj.max()), int((dj > 1e-9).sum()), int(ca.sum())))
