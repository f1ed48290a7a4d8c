"""Size-independent properties at BASELINE sizes (the oracle is too slow there): posterior normalisation,
invariance under permutation / sharding of the loci, determinism, observation-order invariance, and oracle
parity on a random sample drawn from the full-size batch."""
import os
import sys

import numpy as np
import pytest

from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.batch import PileupBatch
from varlociraptor_amd.dist import shard_range

from parity import compare, describe

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


def big_batch(name, n):
    from bench import generate
    return generate(name, n, 0, chunk_loci=25000, workers=8 if n <= 200_000 else 14)


@pytest.fixture(scope="module")
def config3_full():
    cfg = synth.config3()
    b = big_batch("config3", 1_000_000)  # the full size of BASELINE configs[2]
    plan = engine.Plan(cfg.scenario)
    plan.set_max_obs(int(b.depth().sum(axis=1).max()))
    res = plan.call_host(b)
    return cfg, b, plan, res


def test_posteriors_are_normalised_and_finite(config3_full):
    cfg, b, plan, res = config3_full
    assert not (res.status & 0xF).any()
    ps = np.exp(res.ln_posterior)
    assert np.all(np.isfinite(ps))
    assert np.abs(ps.sum(axis=1) - 1.0).max() < 1e-9
    assert np.all((res.map_vaf >= 0.0) & (res.map_vaf <= 1.0))


def test_planted_classes_are_recovered():
    """The generator plants a class per locus (absent / somatic_tumor / germline_het / germline_hom / somatic_normal at 100x):
    the event with the highest posterior must be the planted one for the bulk of every class."""
    cfg = synth.config3()
    b = synth.generate(cfg, 20000, seed=99)
    plan = engine.Plan(cfg.scenario)
    plan.set_max_obs(int(b.depth().sum(axis=1).max()))
    res = plan.call_host(b)
    plan.close()
    names = cfg.scenario.out_names()
    called = np.exp(res.ln_posterior).argmax(axis=1)
    cls, cls_names = b.truth["class"], b.truth["class_names"]
    floor = {"absent": 0.97, "germline_het": 0.90, "germline_hom": 0.90, "somatic_tumor": 0.75, "somatic_normal": 0.40}  # 3 % of the loci carry an injected artifact
    for ci, cname in enumerate(cls_names):
        sel = cls == ci
        assert sel.sum() > 100
        frac = float((called[sel] == names.index(cname)).mean())
        assert frac >= floor[cname], (cname, frac)


def test_sharding_and_permutation_do_not_change_results(config3_full):
    cfg, b, plan, res = config3_full
    n = b.n_loci
    # contiguous shards (the multi-GPU partition) give bit-identical per-locus results
    for rank in (0, 3, 7):
        lo, hi = shard_range(n, rank, 8)
        sub = plan.call_host(b.select(np.arange(lo, hi)))
        assert np.array_equal(sub.ln_posterior, res.ln_posterior[lo:hi])
        assert np.array_equal(sub.map_vaf, res.map_vaf[lo:hi], equal_nan=True)
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)[:50_000]
    sub = plan.call_host(b.select(perm))
    assert np.array_equal(sub.ln_posterior, res.ln_posterior[perm])
    assert np.array_equal(sub.map_bias, res.map_bias[perm])


def test_run_to_run_determinism(config3_full):
    cfg, b, plan, res = config3_full
    again = plan.call_host(b)
    assert np.array_equal(again.ln_posterior, res.ln_posterior)
    assert np.array_equal(again.map_vaf, res.map_vaf, equal_nan=True)
    assert np.array_equal(again.status, res.status)


def test_observation_order_within_a_pileup_is_irrelevant(config3_full):
    """The pileup likelihood is a product over observations; reversing every pileup changes only rounding."""
    cfg, b, plan, res = config3_full
    loci = np.arange(2000)
    sub = b.select(loci)
    rev_cols = {k: v.copy() for k, v in sub.columns.items()}
    for p in range(sub.n_loci * sub.n_samples):
        s, e = int(sub.obs_offset[p]), int(sub.obs_offset[p + 1])
        for k in rev_cols:
            rev_cols[k][s:e] = sub.columns[k][s:e][::-1]
    rev = PileupBatch(sub.n_samples, sub.obs_offset, rev_cols, sub.locus)
    a, c = plan.call_host(sub), plan.call_host(rev)
    d = np.abs(np.exp(a.ln_posterior) - np.exp(c.ln_posterior))
    assert d.max() < 1e-9


def test_random_sample_of_the_full_batch_matches_the_oracle(config3_full, oracle):
    from test_gpu_edge_cases import oracle_mt
    cfg, b, plan, res = config3_full
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(b.n_loci, size=3000, replace=False))
    sub = b.select(pick)
    ref = oracle_mt(oracle, cfg.scenario, sub, threads=min(64, os.cpu_count() or 8))
    from varlociraptor_amd.batch import CallResults
    got = CallResults(len(pick), plan.n_out, plan.n_samples)
    for f in ("ln_posterior", "map_vaf", "map_bias", "best_event", "status"):
        getattr(got, f)[:] = getattr(res, f)[pick]
    m = compare(got, ref, label="config3 sample of the full 1 M")
    print(describe(m))
    assert m["frac_within"] == 1.0, describe(m)


# ---- BASELINE configs[3] and configs[4] at their per-GPU FULL sizes (10 M / 8 and 5 M / 8 loci): the same size-independent
# properties as config 3 above plus oracle parity on a random sample
FULL_SIZES = {"config4": 1_250_000, "config5": 625_000}


@pytest.fixture(scope="module", params=sorted(FULL_SIZES))
def full_config(request):
    name = request.param
    cfg = synth.CONFIGS[name]()
    b = big_batch(name, FULL_SIZES[name])
    plan = engine.Plan(cfg.scenario)
    plan.set_max_obs(min(int(b.depth().sum(axis=1).max()), engine.MAX_OBS_LDS))
    res = plan.call_host(b)
    yield name, cfg, b, plan, res
    plan.close()


def test_full_size_normalisation_determinism_and_shards(full_config):
    name, cfg, b, plan, res = full_config
    n = b.n_loci
    assert n == FULL_SIZES[name]
    assert not (res.status & 0xF).any()
    ps = np.exp(res.ln_posterior)
    assert np.all(np.isfinite(ps))
    assert np.abs(ps.sum(axis=1) - 1.0).max() < 1e-9
    ok = ~np.isnan(res.map_vaf)
    assert np.all((res.map_vaf[ok] >= 0.0) & (res.map_vaf[ok] <= 1.0))
    # the multi-GPU partition: contiguous shards of 8 ranks give bit-identical per-locus results (ranks 0 and 5 here)
    for rank in (0, 5):
        lo, hi = shard_range(n, rank, 8)
        sub = plan.call_host(b.select(np.arange(lo, hi)))
        assert np.array_equal(sub.ln_posterior, res.ln_posterior[lo:hi])
        assert np.array_equal(sub.map_vaf, res.map_vaf[lo:hi], equal_nan=True)
        assert np.array_equal(sub.status, res.status[lo:hi])
    # permutation invariance and run-to-run determinism on a random 60 000-locus subset
    rng = np.random.default_rng(11)
    perm = rng.permutation(n)[:60_000]
    pb = b.select(perm)
    one, two = plan.call_host(pb), plan.call_host(pb)
    for f in ("ln_posterior", "map_bias", "best_event", "status"):
        assert np.array_equal(getattr(one, f), getattr(res, f)[perm]), f
        assert np.array_equal(getattr(one, f), getattr(two, f)), f
    assert np.array_equal(one.map_vaf, res.map_vaf[perm], equal_nan=True)


def test_full_size_random_sample_matches_the_oracle(full_config, oracle):
    from test_gpu_edge_cases import oracle_mt
    from varlociraptor_amd.batch import CallResults
    name, cfg, b, plan, res = full_config
    rng = np.random.default_rng(13)
    pick = np.sort(rng.choice(b.n_loci, size=3000, replace=False))
    ref = oracle_mt(oracle, cfg.scenario, b.select(pick), threads=min(64, os.cpu_count() or 8))
    got = CallResults(len(pick), plan.n_out, plan.n_samples)
    for f in ("ln_posterior", "map_vaf", "map_bias", "best_event", "status"):
        getattr(got, f)[:] = getattr(res, f)[pick]
    m = compare(got, ref, label="%s sample of %d" % (name, b.n_loci))
    print(describe(m))
    assert m["frac_within"] == 1.0, describe(m)
    assert m["status_equal"]


def test_config2_full_size_parity(oracle):
    """BASELINE configs[1] at its full size (100 000 SNV loci, 30x): every locus against the oracle."""
    from test_gpu_edge_cases import oracle_mt
    cfg = synth.config2()
    b = big_batch("config2", 100_000)
    plan = engine.Plan(cfg.scenario)
    got = plan.call_host(b)
    plan.close()
    ref = oracle_mt(oracle, cfg.scenario, b, threads=min(64, os.cpu_count() or 8))
    m = compare(got, ref, label="config2 x100000")
    print(describe(m))
    assert m["frac_within"] == 1.0, describe(m)
    assert m["status_equal"]
    # two artifact hypotheses can explain the alt reads equally well (all alt reads forward AND F1R2): their joint
    # probabilities then differ only by rounding and the reported bias symbol is arbitrary (AF = 0 either way)
    n_label = int((got.map_bias != ref.map_bias).any(axis=1).sum())
    assert n_label <= 5, n_label


def test_breakend_mates_sharing_a_pileup_get_identical_results():
    """config5 carries SV/breakend loci whose mate record shares the pileup (calling.rs:569-580: the reference evaluates
    one of them and reuses the result, 726-741, 820-839); evaluated twice by the engine they must agree bit for bit, and
    observations without strand information (Strand::None) are part of those pileups."""
    import numpy as np
    from varlociraptor_amd import abi, engine, synth
    cfg = synth.config5()
    b = synth.generate(cfg, 3000, seed=77)
    g = b.truth["group"]
    mates = np.nonzero(g != np.arange(b.n_loci))[0]
    assert len(mates) > 20
    strand = (b.columns["flags"] >> abi.F_STRAND_SHIFT) & 3
    assert (strand == abi.STRAND_NONE).sum() > 100
    plan = engine.Plan(cfg.scenario)
    got = plan.call_host(b)
    plan.close()
    for f in ("ln_posterior", "map_vaf", "map_bias", "best_event", "status"):
        a = getattr(got, f)
        assert np.array_equal(a[mates], a[g[mates]], equal_nan=True), f


def test_afd_sub_ranges_on_two_lanes_equal_one_range(oracle, monkeypatch):
    """The AFD log is budgeted (4 GiB): larger batches are walked in sub-ranges of loci on two stream lanes (the caller's stream and a
    plan-owned one).  With a budget of a few MB a 3 000-locus batch takes a dozen sub-ranges: posteriors, MAP and lists must be
    those of the single-range run, bit for bit, and the lists those of the oracle."""
    from varlociraptor_amd import engine, synth
    cfg = synth.config3()
    b = synth.generate(cfg, 3000, seed=33)
    plan = engine.Plan(cfg.scenario)
    one = plan.call_host(b, afd_capacity=96)
    plan.close()
    monkeypatch.setenv("VLR_AFD_LOG_BUDGET_MB", "16")   # ~ 480 loci per log, 240 per lane
    plan = engine.Plan(cfg.scenario)
    many = plan.call_host(b, afd_capacity=96)
    plan.close()
    for f in ("ln_posterior", "map_vaf", "map_bias", "best_event", "status", "afd_count"):
        assert np.array_equal(getattr(one, f), getattr(many, f), equal_nan=True), f
    n = np.minimum(one.afd_count, 96)
    for l in range(0, b.n_loci, 7):
        for s in range(b.n_samples):
            k = int(n[l, s])
            a = sorted(zip(one.afd_vaf[l, s, :k], one.afd_lnprob[l, s, :k]))
            c = sorted(zip(many.afd_vaf[l, s, :k], many.afd_lnprob[l, s, :k]))
            assert a == c, (l, s)
    ref = oracle.call(cfg.scenario, b, afd_capacity=96, begin=0, end=300)
    assert np.array_equal(ref.afd_count[:300], many.afd_count[:300])
