"""Unit identities of the CPU oracle: the reference's own likelihood tests
(src/variants/model/likelihood.rs:273-394) replayed with the reference's test-observation
constructor defaults (src/variants/model/mod.rs:374-402), plus checks of the restated third-party
arithmetic (bio LogProb, itertools_num::linspace, VAFRange::observable_min/max)."""
import ctypes as C
import math
import os

import numpy as np
import pytest
from scipy import integrate, special

from varlociraptor_amd import abi
from varlociraptor_amd.batch import PileupBatch
from varlociraptor_amd.scenario import (Contamination, Inheritance, Sample, Scenario, Species, single_sample, tumor_normal)

NEG_INF = -np.inf
LN05 = math.log(0.5)


def test_observation_batch(specs):
    """model/mod.rs:374-402 `observation(prob_mapping, prob_alt, prob_ref)` for each spec tuple."""
    n = len(specs)
    pm = np.array([s[0] for s in specs], np.float32)
    pa = np.array([s[1] for s in specs], np.float32)
    pr = np.array([s[2] for s in specs], np.float32)
    with np.errstate(divide="ignore"):
        m = np.maximum(pa.astype(np.float64), pr.astype(np.float64))
        missed = m + np.log1p(np.exp(-np.abs(pa.astype(np.float64) - pr.astype(np.float64)))) - math.log(2.0)
    cols = {
        "prob_mapping": pm, "prob_alt": pa, "prob_ref": pr, "prob_missed_allele": missed.astype(np.float32),
        "prob_sample_alt": np.zeros(n, np.float32), "prob_double_overlap": np.zeros(n, np.float32),
        "prob_hit_base": np.full(n, math.log(0.01), np.float32),
        "flags": abi.pack_flags(np.full(n, abi.STRAND_BOTH), np.full(n, abi.ORIENT_NONE), np.zeros(n, bool), np.zeros(n, bool),
                                np.ones(n, bool), np.ones(n, bool), np.full(n, abi.ALTLOCUS_NONE)),
    }
    return PileupBatch(1, np.array([0, n], np.uint32), cols, {})


test_observation_batch.__test__ = False


def test_likelihood_observation_absent_single(oracle):  # likelihood.rs:273-282
    b = test_observation_batch([(0.0, NEG_INF, 0.0)])
    bs = b.as_struct()
    lh = oracle.lib().vlro_lik_obs_single(C.byref(bs), 0, 0.0)
    ref = oracle.lib().vlro_bias_prob_ref_none(C.byref(bs), 0)
    # Artifacts::none().prob_ref = ln .5 + ln .5 + ln(1 - 0.01) + 0 + 0 + ln .5
    assert ref == pytest.approx(3 * LN05 + math.log1p(-math.exp(np.float32(math.log(0.01)))), rel=1e-12)
    assert lh == pytest.approx(ref, rel=1e-12)


def test_likelihood_observation_absent_contaminated(oracle):  # likelihood.rs:284-297
    b = test_observation_batch([(0.0, NEG_INF, 0.0)])
    bs = b.as_struct()
    lh = oracle.lib().vlro_lik_obs_contaminated(C.byref(bs), 0, 1.0, 0.0, 0.0)
    assert lh == pytest.approx(oracle.lib().vlro_bias_prob_ref_none(C.byref(bs), 0), rel=1e-12)


def test_likelihood_pileup_absent(oracle):  # likelihood.rs:299-353
    b = test_observation_batch([(0.0, NEG_INF, 0.0)] * 10)
    bs = b.as_struct()
    L = oracle.lib()
    expect = sum(L.vlro_bias_prob_ref_none(C.byref(bs), i) for i in range(10))
    assert L.vlro_pileup_lik_contaminated(C.byref(bs), 0, 10, 1.0, 0.0, 0.0) == pytest.approx(expect, rel=1e-12)
    assert L.vlro_pileup_lik_single(C.byref(bs), 0, 10, 0.0) == pytest.approx(expect, rel=1e-12)


def test_likelihood_pileup_max_at_half(oracle):  # likelihood.rs:355-394
    b = test_observation_batch([(0.0, 0.0, NEG_INF)] * 5 + [(0.0, NEG_INF, 0.0)] * 5)
    bs = b.as_struct()
    L = oracle.lib()
    lh = L.vlro_pileup_lik_contaminated(C.byref(bs), 0, 10, 1.0, 0.5, 0.0)
    for af in np.linspace(0.0, 1.0, 10):
        if af != 0.5:
            assert lh > L.vlro_pileup_lik_contaminated(C.byref(bs), 0, 10, 1.0, float(af), 0.0)


def test_logprob_primitives(oracle):
    L = oracle.lib()
    for p in [-1e-9, -0.1, -0.693, -0.6931471805599453, -0.7, -5.0, -50.0, -800.0]:
        assert L.vlro_ln_one_minus_exp(p) == pytest.approx(math.log(-math.expm1(p)), rel=1e-13)
    assert L.vlro_ln_one_minus_exp(NEG_INF) == 0.0
    assert L.vlro_ln_one_minus_exp(0.0) == NEG_INF
    assert L.vlro_ln_add_exp(NEG_INF, NEG_INF) == NEG_INF
    assert L.vlro_ln_add_exp(-3.0, NEG_INF) == -3.0
    rng = np.random.default_rng(1)
    for _ in range(20):
        v = rng.uniform(-50, 0, size=rng.integers(1, 9))
        arr = (C.c_double * len(v))(*v)
        assert L.vlro_ln_sum_exp(arr, len(v)) == pytest.approx(special.logsumexp(v), rel=1e-13)
    arr = (C.c_double * 3)(NEG_INF, NEG_INF, NEG_INF)
    assert L.vlro_ln_sum_exp(arr, 3) == NEG_INF
    assert L.vlro_ln_sum_exp(arr, 0) == NEG_INF


def test_observable_bounds(oracle):
    L = oracle.lib()
    # ]0,1] at n = 85 -> 1/85 (the fixture's first visited point 0.012)
    assert L.vlro_observable_min(0.0, 1.0, 1, 0, 85) == 1.0 / 85.0
    assert L.vlro_observable_max(0.0, 1.0, 1, 0, 85) == 1.0
    # n < 10: no adjustment, even for an exclusive bound (formula.rs:1171-1173)
    assert L.vlro_observable_min(0.0, 1.0, 1, 0, 9) == 0.0
    # ]0,0.5[ at n = 100 -> [1/100, 49/100]
    assert L.vlro_observable_min(0.0, 0.5, 1, 1, 100) == 0.01
    assert L.vlro_observable_max(0.0, 0.5, 1, 1, 100) == 0.49
    # odd n: floor(0.5 n)/n
    assert L.vlro_observable_max(0.0, 0.5, 1, 1, 101) == 50.0 / 101.0
    # inclusive bounds are kept when they are multiples of 1/n
    assert L.vlro_observable_min(0.5, 1.0, 0, 0, 100) == 0.5
    # adjustment impossible when n * width <= 1
    assert L.vlro_observable_max(0.2, 0.205, 0, 1, 100) == 0.205


def test_adaptive_integration_of_a_gaussian(oracle):
    """utils/adaptive_integration.rs on a unimodal density: trapezoid over the visited points is close to
    the true integral and the point count matches the 57/27 figures of SURVEY §8."""
    L = oracle.lib()
    n = C.c_int()
    mu, sigma = 0.3, 0.05
    got = L.vlro_adaptive_gauss(0.01, 1.0, 0.01, mu, sigma, C.byref(n))
    true = math.log(integrate.quad(lambda x: math.exp(-(x - mu) ** 2 / (2 * sigma ** 2)), 0.01, 1.0)[0])
    assert abs(got - true) < 0.05
    assert 50 <= n.value <= 60
    got = L.vlro_adaptive_gauss(0.01, 0.49, 0.1, 0.2, 0.2, C.byref(n))
    assert 20 <= n.value <= 30


def pedigree_scenario():
    """tests/resources/prior/scenarios/pedigree.scenario.yaml shape: ploidy-derived
    universes {0, .5, 1}, Mendelian inheritance from two parents."""
    species = Species(heterozygosity=0.001, germline_mutation_rate=1e-3, ploidy=2)
    samples = {
        "mother": Sample(resolution=0.1), "father": Sample(resolution=0.1),
        "child": Sample(resolution=0.1, inheritance=Inheritance(abi.INHERIT_MENDELIAN, ("mother", "father"))),
        "sibling": Sample(resolution=0.1, inheritance=Inheritance(abi.INHERIT_MENDELIAN, ("mother", "father"))),
    }
    events = {
        "denovo_child": "child:0.5 & mother:0.0 & father:0.0 & sibling:0.0",
        "inherited": "(mother:0.5 | mother:1.0 | father:0.5 | father:1.0)",
    }
    return Scenario(samples, events, species=species)


def test_mendelian_prior(oracle):
    sc = pedigree_scenario()
    assert sc.sample_names == ["child", "father", "mother", "sibling"]
    d = sc.desc()
    L = oracle.lib()

    def prior(child, father, mother, sibling, vt=abi.VT_SNV):
        v = (C.c_double * 4)(child, father, mother, sibling)
        return L.vlro_prior(C.byref(d), v, vt)

    # absent-only mode (calling.rs:1086): every possible non-absent tuple gets ln(1 - P(all absent))
    p_abs = prior(0, 0, 0, 0)
    p_non = math.log(-math.expm1(p_abs))
    assert prior(0.5, 0.5, 0.0, 0.0) == pytest.approx(p_non, rel=1e-12)
    assert prior(0.5, 0.0, 0.0, 0.0) == pytest.approx(p_non, rel=1e-12)  # de novo: possible via germline mutation
    assert prior(0.0, 1.0, 1.0, 0.0) == pytest.approx(p_non, rel=1e-12) or prior(0.0, 1.0, 1.0, 0.0) == NEG_INF
    assert prior(0.25, 0.0, 0.0, 0.0) == NEG_INF  # not a valid germline VAF for ploidy 2
    # full prior: heterozygosity term for the founders, hypergeometric inheritance for the children
    sc2 = pedigree_scenario()
    sc2.full_prior = True
    d2 = sc2.desc()

    def fprior(*v):
        arr = (C.c_double * 4)(*v)
        return L.vlro_prior(C.byref(d2), arr, abi.VT_SNV)

    het = math.log(0.001)
    # one alt allele in the father (m = 1): prior.rs:554-582 -> heterozygosity / 1; each child inherits ref with prob 1/2
    assert fprior(0.0, 0.5, 0.0, 0.0) == pytest.approx(het + 2 * math.log(0.5), rel=1e-9)
    # child het from het father: alt inherited (1/2) or ref inherited (1/2) + one germline mutation (1e-3)
    assert fprior(0.5, 0.5, 0.0, 0.0) == pytest.approx(het + math.log(0.5 * (1 + 1e-3)) + math.log(0.5), rel=1e-9)
    # de novo in the child only: no alt in founders -> 1 - sum_m het/m ; child needs one germline mutation (rate 1e-3)
    p0 = math.log(1.0 - sum(0.001 / m for m in range(1, 5)))
    assert fprior(0.5, 0.0, 0.0, 0.0) == pytest.approx(p0 + math.log(1e-3), rel=1e-9)


def test_clonal_inheritance_of_somatic_vaf_without_own_rate(oracle):
    """prior.rs:489-499 (`(true, None)` arm of prob_clonal_inheritance): a sample that inherits clonally, somatic VAF
    included, and has no somatic rate of its own must carry exactly the parent's VAF, which then has to be one of the
    germline levels (the sample has no somatic variation, calc_prob pins its germline VAF to its VAF, prior.rs:417-427)."""
    import ctypes as C
    from varlociraptor_amd.scenario import Inheritance, Sample, Scenario, Species
    species = Species(heterozygosity=0.001, germline_mutation_rate=1e-3, ploidy=2, somatic_effective_mutation_rate=None)
    sc = Scenario({"p": Sample(somatic_effective_mutation_rate=1e-6, resolution=0.1),
                   "r": Sample(resolution=0.1, inheritance=Inheritance(abi.INHERIT_CLONAL, ("p",), True))},
                  {"het": "p:0.5 & r:0.5", "hom": "p:1.0 & r:1.0", "sub": "p:]0.0,0.5["}, species=species, full_prior=True)
    d = sc.desc()
    L = oracle.lib()

    def prior(p, r):
        return L.vlro_prior(C.byref(d), (C.c_double * 2)(p, r), abi.VT_SNV)

    assert prior(0.5, 0.5) > NEG_INF and prior(1.0, 1.0) > NEG_INF and prior(0.0, 0.0) > NEG_INF
    assert prior(0.5, 1.0) == NEG_INF          # germline VAFs differ
    assert prior(0.3, 0.3) == NEG_INF          # 0.3 is not a germline level of the relapse sample
    assert prior(0.3, 0.0) == NEG_INF          # parent carries a somatic VAF the child cannot have inherited unchanged
    assert prior(0.3, 0.5) == NEG_INF


def test_tuned_cpu_baseline_agrees_with_the_fidelity_path(oracle):
    """bench.py's cpu_baseline.tuned: the affine product form of the pileup likelihood against the log-space
    restatement on every BASELINE workload shape, incl. loci whose terms leave the linear range (those keep the log-space code)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity import compare, describe
    from varlociraptor_amd import synth
    for name, n in (("config2", 300), ("config3", 60), ("config4", 40), ("config5", 120)):
        cfg = synth.CONFIGS[name]()
        b = synth.generate(cfg, n, seed=19)
        ref = oracle.call(cfg.scenario, b, want_events=True)
        got = oracle.call(cfg.scenario, b, want_events=True, tuned=True)
        m = compare(got, ref, label="tuned " + name)
        assert m["frac_within"] == 1.0 and m["bias_equal"] and m["status_equal"], describe(m)
    cfg = synth.config3()
    b = synth.generate(cfg, 30, seed=23)
    b.columns["prob_alt"][::3] = -740.0   # supports below the f64 linear range
    ref = oracle.call(cfg.scenario, b, want_events=True)
    got = oracle.call(cfg.scenario, b, want_events=True, tuned=True)
    m = compare(got, ref, label="tuned deep supports")
    assert m["frac_within"] == 1.0, describe(m)
