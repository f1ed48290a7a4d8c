"""Shared comparison helpers for the parity tests (GPU engine vs CPU oracle)."""
import numpy as np

TOL = 1e-6  # BASELINE.json north_star: posteriors and MAP allele frequencies within 1e-6 absolute


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
    bad = np.nonzero(per_locus > tol)[0]
    return {
        "label": label,
        "n": len(per_locus),
        "max_dpost": float(np.nan_to_num(dp, nan=np.inf).max()) if dp.size else 0.0,
        "max_dvaf": float(np.nan_to_num(dv, nan=np.inf).max()) if dv.size else 0.0,  # exact-tie loci excluded
        "frac_within": float((per_locus <= tol).mean()) if len(per_locus) else 1.0,
        "bad": bad,
        "bias_equal": bool((got.map_bias[~ties] == ref.map_bias[~ties]).all()),
        "best_equal_frac": float((got.best_event == ref.best_event)[~ties].mean()) if (~ties).any() else 1.0,
        "n_ties": int(ties.sum()),
        "status_equal": bool((got.status == ref.status).all()),
    }


def describe(m):
    return ("%s: n=%d max|dpost|=%.3g max|dvaf|=%.3g within=%.6f bias_equal=%s best_equal=%.4f status_equal=%s ties=%d bad=%s" %
            (m["label"], m["n"], m["max_dpost"], m["max_dvaf"], m["frac_within"], m["bias_equal"], m["best_equal_frac"],
             m["status_equal"], m["n_ties"], [int(x) for x in m["bad"][:10]]))
