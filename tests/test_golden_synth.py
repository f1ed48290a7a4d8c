"""Committed golden vectors of the oracle on seeded synthetic loci (tests/golden/synth/*.npz, made by
tools/make_golden_synth.py): the oracle must still produce them (CPU suite — a silent change of the restatement is caught
even though every parity test compares engine and oracle with each other), and the engine must match them on the GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CONFIGS = ["config2", "config3", "config4", "config5"]


def _load(golden_dir, name):
    import make_golden_synth as mg
    g = np.load(os.path.join(golden_dir, "synth", name + ".npz"))
    cfg, batch = mg.inputs(name)
    assert mg.digest(batch) == str(g["input_digest"][0]), "the synthetic generator no longer reproduces the golden inputs"
    return mg, g, cfg, batch


def _close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore"):
        d = np.where(same, 0.0, np.abs(a - b))
    return float(np.nan_to_num(d, nan=np.inf).max()) <= tol if d.size else True


@pytest.mark.parametrize("name", CONFIGS)
def test_oracle_reproduces_its_golden_vectors(oracle, golden_dir, name):
    mg, g, cfg, batch = _load(golden_dir, name)
    out = mg.evaluate(cfg, batch)
    assert list(g["out_names"]) == cfg.scenario.out_names()
    for f in ("map_bias", "best_event", "status", "afd_count"):
        assert np.array_equal(out[f], g[f]), f
    assert np.array_equal(out["map_vaf"], g["map_vaf"], equal_nan=True)
    assert np.array_equal(out["afd_vaf"], g["afd_vaf"])           # the visited points are exact
    for f in ("ln_posterior", "ln_marginal", "event_ln_posterior", "afd_lnprob"):
        assert _close(out[f], g[f], 1e-9), f                      # libm may differ in the last bits between machines


@pytest.mark.gpu
@pytest.mark.parametrize("name", CONFIGS)
def test_engine_matches_golden_vectors(golden_dir, name):
    from varlociraptor_amd import engine
    from varlociraptor_amd.batch import CallResults
    from parity import compare, describe
    mg, g, cfg, batch = _load(golden_dir, name)
    plan = engine.Plan(cfg.scenario)
    plan.set_max_obs(int(batch.depth().sum(axis=1).max()))
    got = plan.call_host(batch)
    afd = plan.call_host(batch.select(np.arange(mg.N_AFD)), afd_capacity=mg.AFD_CAP)
    plan.close()
    ref = CallResults(batch.n_loci, cfg.scenario.n_out, batch.n_samples)
    for f in ("ln_posterior", "ln_marginal", "map_vaf", "map_bias", "best_event", "status"):
        getattr(ref, f)[:] = g[f]
    ref.event_ln_posterior = g["event_ln_posterior"]
    m = compare(got, ref, label=name)
    print(describe(m))
    assert m["frac_within"] == 1.0 and m["bias_equal"] and m["status_equal"], describe(m)
    # visited-point lists: same points (as sets per sample), densities within 1e-6
    ties = 0
    for l in range(mg.N_AFD):
        if got.best_event[l] != g["best_event"][l]:
            ties += 1  # exact event tie (counted by compare): the AFD follows the chosen event
            continue
        for s in range(batch.n_samples):
            ng, nr = int(afd.afd_count[l, s]), int(g["afd_count"][l, s])
            assert ng == nr, (l, s, ng, nr)
            og, orr = np.argsort(afd.afd_vaf[l, s, :ng], kind="stable"), np.argsort(g["afd_vaf"][l, s, :nr], kind="stable")
            assert np.array_equal(afd.afd_vaf[l, s, :ng][og], g["afd_vaf"][l, s, :nr][orr]), (l, s)
            assert np.allclose(np.sort(afd.afd_lnprob[l, s, :ng]), np.sort(g["afd_lnprob"][l, s, :nr]), atol=1e-6, equal_nan=True), (l, s)
    assert ties <= m["n_ties"]
