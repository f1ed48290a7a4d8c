"""Shared comparison helpers for the parity tests (GPU engine vs CPU oracle)."""
import numpy as np

TOL = 1e-6  # BASELINE.json north_star: posteriors and MAP allele frequencies within 1e-6 absolute
# The product emits posteriors as PHRED-scaled f32 (PROB_* of the calls record, calling/variants/mod.rs:447-559: 6-7 significant
# digits of -10 log10 p) and the reference's testcases assert on those (testcase/runner/common/mod.rs:332-393): an event whose
# posterior is 1e-30 must agree in LOG space too, not only to 1e-6 absolute.  Criterion per entry, finite on both sides:
# |d ln p| <= LN_TOL * max(1, |ln p|), i.e. PHRED values equal to >= 6 digits; -inf only where the reference has -inf.
LN_TOL = 1e-6


def ln_deviation(lg, lr):
    """(relative log-space deviation per entry, entries that break the log-space criterion outright: -inf / NaN on one side only)."""
    lg, lr = np.asarray(lg, np.float64), np.asarray(lr, np.float64)
    fin = np.isfinite(lg) & np.isfinite(lr)
    with np.errstate(invalid="ignore"):
        rel = np.where(fin, np.abs(lg - lr) / np.maximum(1.0, np.abs(lr)), 0.0)
    same_special = (np.isnan(lg) & np.isnan(lr)) | (np.isneginf(lg) & np.isneginf(lr)) | (np.isposinf(lg) & np.isposinf(lr))
    broken = ~fin & ~same_special
    return rel, broken


def compare(got, ref, tol=TOL, label="", tie_tol=1e-9):
    """Return a dict of parity metrics; posteriors compared as probabilities (exp of ln posterior)."""
    pg, pr = np.exp(got.ln_posterior), np.exp(ref.ln_posterior)
    dp = np.abs(pg - pr)
    dp = np.where(np.isnan(pg) & np.isnan(pr), 0.0, dp)
    dv = np.abs(got.map_vaf - ref.map_vaf)
    dv = np.where(np.isnan(got.map_vaf) & np.isnan(ref.map_vaf), 0.0, dv)
    # Exact ties between event posteriors (e.g. a single, singleton-adjusted observation makes the likelihood
    # flat, so `absent` and `present` tie) are broken by rounding noise: the reference's own choice there is
    # arbitrary (SURVEY §7 hard part 3).  Such loci are counted separately; their MAP follows the chosen event.
    ties = np.zeros(len(dp), bool)
    ev = getattr(ref, "event_ln_posterior", None)
    if ev is not None and len(dp):
        idx = np.arange(len(dp))
        differ = got.best_event != ref.best_event
        with np.errstate(invalid="ignore"):
            gap = np.abs(ev[idx, np.clip(got.best_event, 0, ev.shape[1] - 1)] - ev[idx, np.clip(ref.best_event, 0, ev.shape[1] - 1)])
        ties = differ & (gap < tie_tol)
        dv = np.where(ties[:, None], 0.0, dv)
    per_locus = np.maximum(np.nan_to_num(dp, nan=np.inf).max(axis=1), np.nan_to_num(dv, nan=np.inf).max(axis=1))
    # log space: what the calls record carries (PHRED f32)
    rel, broken = ln_deviation(got.ln_posterior, ref.ln_posterior)
    ln_fail = (rel > LN_TOL).any(axis=1) | broken.any(axis=1) if rel.size else np.zeros(len(per_locus), bool)
    per_locus = np.where(ln_fail, np.inf, per_locus)
    bad = np.nonzero(per_locus > tol)[0]
    return {
        "label": label,
        "n": len(per_locus),
        "max_dpost": float(np.nan_to_num(dp, nan=np.inf).max()) if dp.size else 0.0,
        "max_dvaf": float(np.nan_to_num(dv, nan=np.inf).max()) if dv.size else 0.0,  # exact-tie loci excluded
        "max_dln": float(rel.max()) if rel.size else 0.0,   # max |d ln posterior| / max(1, |ln posterior|) over entries finite on both sides
        "n_ln_broken": int(broken.sum()),                   # entries that are -inf / NaN on one side only
        "n_ln_fail": int(ln_fail.sum()),                    # loci that break the log-space criterion (they are in `bad`)
        "frac_within": float((per_locus <= tol).mean()) if len(per_locus) else 1.0,
        "bad": bad,
        "bias_equal": bool((got.map_bias[~ties] == ref.map_bias[~ties]).all()),
        "best_equal_frac": float((got.best_event == ref.best_event)[~ties].mean()) if (~ties).any() else 1.0,
        "n_ties": int(ties.sum()),
        "status_equal": bool((got.status == ref.status).all()),
    }


def describe(m):
    return ("%s: n=%d max|dpost|=%.3g max_dln=%.3g (broken %d) max|dvaf|=%.3g within=%.6f bias_equal=%s best_equal=%.4f status_equal=%s ties=%d bad=%s" %
            (m["label"], m["n"], m["max_dpost"], m["max_dln"], m["n_ln_broken"], m["max_dvaf"], m["frac_within"], m["bias_equal"], m["best_equal_frac"],
             m["status_equal"], m["n_ties"], [int(x) for x in m["bad"][:10]]))
