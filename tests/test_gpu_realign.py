"""GPU parity of the pair-HMM kernel (vlr_realign_batch, varlociraptor_amd/csrc/vlr_realign.hip) against the CPU restatement
(oracle/vlr_realign_oracle.cpp) through the C ABI: random read/allele windows of every variant kind, banded and unbanded,
with and without gap extension, edge shapes (one base, 128-base reads, long alleles, lower case, empty, oversize), the
underflow guard, and the size-independent property that ref/alt supports normalise to one.  Tolerance: 1e-9 absolute on
ln P (the kernel multiplies in linear space, the oracle adds logarithms); BASELINE's 1e-6 bar applies to the normalised
probabilities, checked as well."""
import math

import numpy as np
import pytest

from varlociraptor_amd import realign, realign_synth
from varlociraptor_amd.realign import GapParams, PairBatch

pytestmark = pytest.mark.gpu
TOL = 1e-9


def check(oracle, pb, gap=None, tol=TOL):
    gap = gap or GapParams()
    got = realign.prob_related(pb, gap)
    ref = oracle.pairhmm_batch(pb, gap, threads=8)
    both_inf = np.isneginf(got) & np.isneginf(ref)
    d = np.where(both_inf, 0.0, np.abs(got - ref))
    assert np.all(d <= tol * np.maximum(1.0, np.abs(ref) * 1e-3)), (float(np.nanmax(d)), int(np.nanargmax(d)))
    return got, ref


@pytest.mark.parametrize("banded", [True, False])
def test_random_windows_match_oracle(oracle, banded):
    pb, truth = realign_synth.generate(300, seed=11, banded=banded)
    got, ref = check(oracle, pb)
    # normalised supports (what the observation records carry) within BASELINE's 1e-6
    for k in range(len(truth)):
        g = realign.normalize_support(got[2 * k], got[2 * k + 1])
        r = oracle.normalize_support(ref[2 * k], ref[2 * k + 1])
        assert abs(math.exp(g[0]) - math.exp(r[0])) <= 1e-6 and abs(math.exp(g[1]) - math.exp(r[1])) <= 1e-6
        assert abs(math.exp(g[0]) + math.exp(g[1]) - 1.0) < 1e-9
    # reads drawn from the alt allele support it
    sup = np.array([got[2 * k + 1] > got[2 * k] for k in range(len(truth))])
    assert (sup == truth).mean() > 0.9


def test_gap_extension(oracle):
    pb, _ = realign_synth.generate(120, seed=12)
    check(oracle, pb, GapParams(math.log(1e-4), math.log(2e-4), math.log(0.2), math.log(0.3)))


def test_edge_shapes(oracle):
    rng = np.random.default_rng(5)
    B = np.frombuffer(b"ACGT", np.uint8)
    pb = PairBatch()
    pb.add(b"A", b"A", [30])
    pb.add(b"C", b"A", [30])
    pb.add(b"ACGT", b"A", [40])
    pb.add(b"AAAA", b"A", [40])                       # capped at ln 1
    x = B[rng.integers(0, 4, 300)].tobytes()
    pb.add(x, x[100:228], [37] * 128)                 # the longest read window
    pb.add(x.lower(), x[100:228], [37] * 128)         # lower-case reference
    pb.add(x[:3], x[:64], [20] * 64)                  # read much longer than the allele
    pb.add(B[rng.integers(0, 4, 2000)].tobytes(), x[10:90], [30] * 80)   # long allele window
    y = B[rng.integers(0, 4, 128)].tobytes()
    pb.add(x[:200], y, [40] * 128)                    # unrelated read at Q40: ~1e-400, exercises the rescaling
    pb.add(x[:200], y, [93] * 128)                    # and at Q93
    pb.add(x[:200], x[20:120], [30] * 100, max_edit_dist=0)   # tightest band
    pb.add(x[:200], y[:100], [30] * 100, max_edit_dist=3)     # band that excludes every path
    got, ref = check(oracle, pb)
    assert got[3] == 0.0
    assert got[8] < -700 and got[9] < -1300   # far below the f64 range in linear space
    assert np.isneginf(got[11]) and np.isneginf(ref[11])


def test_empty_and_oversize_inputs():
    pb = PairBatch()
    assert len(realign.prob_related(pb)) == 0
    pb.add(b"ACGT", b"", [])
    pb.add(b"", b"ACGT", [30] * 4)
    pb.add(b"A" * 300, b"A" * 129, [30] * 129)
    got = realign.prob_related(pb)
    assert np.isneginf(got[0]) and np.isneginf(got[1]) and np.isnan(got[2])


def test_allele_support_pipeline(oracle):
    """allele_support(): pre-filter + banded kernel + normalisation for a deletion, checked against the oracle path."""
    rng = np.random.default_rng(9)
    locus = realign_synth.make_locus(rng, kind="del")
    reads = [realign_synth.make_read(rng, locus, from_alt=bool(k % 2)) for k in range(40)]
    sup = realign.allele_support(reads, locus["ref_allele"], locus["alt_allele"])
    for k, (seq, q) in enumerate(reads):
        pr = oracle.pairhmm_prob_related(locus["ref_allele"], seq, q, [math.log(2.8e-6), math.log(5.1e-6), -math.inf, -math.inf],
                                         realign.best_hit(seq, locus["ref_allele"])[0] + realign.EDIT_BAND)
        pa = oracle.pairhmm_prob_related(locus["alt_allele"], seq, q, [math.log(2.8e-6), math.log(5.1e-6), -math.inf, -math.inf],
                                         realign.best_hit(seq, locus["alt_allele"])[0] + realign.EDIT_BAND)
        r, a = oracle.normalize_support(pr, pa)
        assert abs(math.exp(sup[k, 0]) - math.exp(r)) <= 1e-6 and abs(math.exp(sup[k, 1]) - math.exp(a)) <= 1e-6


def test_edit_distance_kernel_is_bit_exact(oracle):
    """vlr_edit_distance_batch (integers: bit-exact) against the CPU restatement on the synthetic realignment windows, on random
    pairs of every shape (one base to 128-base reads, alleles shorter than the read, lower case) and on degenerate input."""
    pb, _ = realign_synth.generate(300, seed=21, banded=False)
    rng = np.random.default_rng(8)
    B = np.frombuffer(b"ACGTacgtN", np.uint8)
    for _ in range(400):
        ly = int(rng.choice([1, 2, 3, 7, 31, 63, 64, 65, 100, 127, 128]))
        lx = int(rng.choice([1, 2, 5, 40, 128, 129, 192, 300]))
        x = B[rng.integers(0, len(B), lx)].tobytes()
        if rng.random() < 0.5 and lx > ly:  # the read really comes from the allele, with a few edits
            s = int(rng.integers(0, lx - ly + 1))
            y = bytearray(x[s:s + ly].upper())
            for _ in range(int(rng.integers(0, 4))):
                y[int(rng.integers(0, ly))] = B[int(rng.integers(0, 4))]
            y = bytes(y)
        else:
            y = B[rng.integers(0, 4, ly)].tobytes()
        pb.add(x, y, [30] * ly)
    pb.add(b"", b"A", [30])
    pb.add(b"A", b"", [])
    pb.add(b"ACGT" * 40, b"A" * 129, [30] * 129)
    dist, end, hits = realign.best_hits(pb)
    for k in range(len(pb)):
        assert (int(dist[k]), int(end[k]), int(hits[k])) == oracle.edit_distance(pb.x[k], pb.y[k]), k
    # the band the pipeline derives from it is the one the host routine gives
    for k in range(0, 200, 7):
        h = realign.best_hit(pb.y[k], pb.x[k])
        assert h is not None and h[0] == dist[k] and h[1] == end[k]


def test_resident_pairs_take_their_band_from_the_edit_distance_kernel(oracle):
    import torch
    pb, _ = realign_synth.generate(64, seed=23, banded=False)
    dp = realign.DevicePairs(pb)
    dist = dp.band_from_hits().cpu().numpy()
    torch.cuda.synchronize()
    want = np.array([oracle.edit_distance(pb.x[k], pb.y[k])[0] for k in range(len(pb))])
    assert np.array_equal(dist, want)
    got = dp.run().cpu().numpy()
    pb.band = [int(d) + realign.EDIT_BAND for d in want]
    ref = oracle.pairhmm_batch(pb, GapParams(), threads=8)
    assert np.all(np.abs(got - ref) <= TOL * np.maximum(1.0, np.abs(ref) * 1e-3))


def test_fast_mode_matches_restatement(oracle):
    """vlr_realign_fast_batch (PathHMMRealigner, realignment/mod.rs:547-678): best path probability over the alignments of
    minimal edit distance, max-plus wavefront in log space, against the CPU restatement (itself checked against brute-force
    enumeration in tests/test_realign_oracle.py).  Random windows of every variant kind, gap extension, edge shapes."""
    for gap in (GapParams(), GapParams(math.log(1e-4), math.log(2e-4), math.log(0.2), math.log(0.3))):
        pb, truth = realign_synth.generate(250, seed=13, banded=False)
        g = [gap.prob_insertion_artifact, gap.prob_deletion_artifact, gap.prob_insertion_extend_artifact, gap.prob_deletion_extend_artifact]
        got = realign.prob_best_path(pb, gap)
        ref = np.array([oracle.pathhmm_best(pb.x[k], pb.y[k], pb.q[k], g) for k in range(len(pb))])
        d = np.where(np.isneginf(got) & np.isneginf(ref), 0.0, np.abs(got - ref))
        assert np.all(d <= 1e-9 * np.maximum(1.0, np.abs(ref))), (float(np.nanmax(d)), int(np.nanargmax(d)))
        sup = np.array([got[2 * k + 1] > got[2 * k] for k in range(len(truth))])
        assert (sup == truth).mean() > 0.9
    rng = np.random.default_rng(6)
    B = np.frombuffer(b"ACGT", np.uint8)
    pb = PairBatch()
    pb.add(b"A", b"A", [30]); pb.add(b"C", b"A", [30]); pb.add(b"ACGT", b"A", [40]); pb.add(b"AAAA", b"A", [40])
    x = B[rng.integers(0, 4, 300)].tobytes()
    pb.add(x, x[100:228], [37] * 128)
    pb.add(x.lower(), x[100:228], [37] * 128)
    pb.add(x[:3], x[:64], [20] * 64)                   # read much longer than the allele: leading / trailing insertions
    pb.add(x, x[10:40] + x[45:90], [30] * 75)          # a deletion in the read
    pb.add(x, x[10:40] + b"ACGTT" + x[40:90], [30] * 85)  # an insertion
    for ly in (1, 2, 3, 63, 64, 65, 127, 128):
        pb.add(x, x[50:50 + ly], [35] * ly)
    g = GapParams()
    gl = [g.prob_insertion_artifact, g.prob_deletion_artifact, g.prob_insertion_extend_artifact, g.prob_deletion_extend_artifact]
    got = realign.prob_best_path(pb, g)
    ref = np.array([oracle.pathhmm_best(pb.x[k], pb.y[k], pb.q[k], gl) for k in range(len(pb))])
    assert np.allclose(got, ref, rtol=0, atol=1e-9), (got, ref)


def test_fast_mode_bound_vs_concrete_traceback_stays_below_the_recorded_gap(oracle):
    """The `fast` kernel is an UPPER bound over all co-optimal alignments (include/vlr.h, vlr_realign_fast_batch); the reference
    evaluates the one alignment bio's Myers traceback returns (realignment/mod.rs:547-678).  On the workload of
    tools/fast_mode_gap.py (profiles/r04_fast_mode_gap.json: 600 reads, seed 7) the kernel must never fall below a concrete
    diagonal-first traceback, and the NORMALISED ref/alt supports — what the observation records carry — must not move by more
    than the recorded maximum (1.4e-13, asserted with a decade of slack and far inside BASELINE's 1e-6)."""
    import json, os
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "..", "profiles", "r04_fast_mode_gap.json")))
    pb, truth = realign_synth.generate(rec["reads"], seed=7)
    assert len(pb) == rec["pairs"]
    gap = GapParams()
    g = [gap.prob_insertion_artifact, gap.prob_deletion_artifact, gap.prob_insertion_extend_artifact, gap.prob_deletion_extend_artifact]
    best = realign.prob_best_path(pb, gap)
    fixed = np.array([oracle.pathhmm_fixed_traceback(pb.x[k], pb.y[k], pb.q[k], g)[0] for k in range(len(pb))])
    gap_ln = best - fixed
    assert (gap_ln > -1e-9 * np.maximum(1.0, np.abs(fixed))).all(), "the maximum over all optimal alignments cannot be below one of them"
    assert gap_ln.max() <= rec["max_gap_ln_p"] + 1e-6                      # ln P itself: up to 348 on unrelated windows
    frac_open = float((gap_ln > 1e-9).mean())
    assert abs(frac_open - rec["frac_pairs_where_the_bound_is_not_attained_by_the_fixed_traceback"]) < 0.01
    moved = 0.0
    for k in range(len(truth)):
        b = realign.normalize_support(best[2 * k], best[2 * k + 1]); f = realign.normalize_support(fixed[2 * k], fixed[2 * k + 1])
        moved = max(moved, abs(math.exp(b[0]) - math.exp(f[0])), abs(math.exp(b[1]) - math.exp(f[1])))
    assert moved <= max(10.0 * rec["max_abs_d_normalised_support"], 1e-12), moved


def test_two_pairs_per_wave_on_short_read_windows(oracle):
    """vlr_realign_kernel2: read windows of at most 64 bases run two pairs per wave (lanes 0-31 / 32-63).  Short windows of
    every variant kind, banded and not, odd pair counts, and batches that mix short and long windows pair by pair."""
    for banded in (True, False):
        pb, truth = realign_synth.generate(201, seed=17, window=24, banded=banded)   # read windows of 24..48 bases
        assert max(len(y) for y in pb.y) <= 64
        check(oracle, pb)
    rng = np.random.default_rng(8)
    B = np.frombuffer(b"ACGT", np.uint8)
    x = B[rng.integers(0, 4, 260)].tobytes()
    pb = PairBatch()
    for ly in (1, 2, 31, 32, 33, 63, 64, 65, 100, 128, 64, 5, 64, 64, 3):   # neighbours (2w, 2w+1): short+short, short+long, long+long, odd tail
        o = int(rng.integers(0, 100))
        pb.add(x, x[o:o + ly], [int(q) for q in rng.choice([20, 30, 40], ly)], int(rng.choice([-1, 4, 9])))
    check(oracle, pb, GapParams(math.log(1e-4), math.log(2e-4), math.log(0.2), math.log(0.3)))


# ---- homopolymer mode (vlr_realign_homopolymer_batch; HomopolyPairHMMRealigner, realignment/mod.rs:680-730) ----------------

def _hop(rng):
    from varlociraptor_amd.realign import HopParams
    return HopParams([math.log(v) for v in rng.uniform(0.001, 0.05, 4)], [math.log(v) for v in rng.uniform(0.001, 0.05, 4)],
                     [math.log(v) for v in rng.uniform(0.05, 0.5, 4)], [math.log(v) for v in rng.uniform(0.05, 0.5, 4)])


def _check_homopoly(oracle, pb, gap, hop, tol=TOL):
    got = realign.prob_related_homopolymer(pb, gap, hop)
    ref = oracle.homopoly_batch(pb, gap, hop)
    both_inf = np.isneginf(got) & np.isneginf(ref)
    d = np.where(both_inf, 0.0, np.abs(got - ref))
    assert np.all(d <= tol * np.maximum(1.0, np.abs(ref) * 1e-3)), (float(np.nanmax(d)), int(np.nanargmax(d)))
    return got, ref


@pytest.mark.parametrize("banded", [True, False])
def test_homopolymer_mode_matches_restatement(oracle, banded):
    rng = np.random.default_rng(31)
    pb, _ = realign_synth.generate(150, seed=13, banded=banded)
    _check_homopoly(oracle, pb, GapParams(), _hop(rng))
    _check_homopoly(oracle, pb, GapParams(math.log(1e-4), math.log(2e-4), math.log(0.2), math.log(0.3)), _hop(rng))
    # homopolymer-rich windows (nanopore-like run length errors), lower case and N bases in the allele
    B = np.frombuffer(b"AAAACCGTTTT", np.uint8)
    pb2 = PairBatch()
    for k in range(200):
        x = bytearray(B[rng.integers(0, len(B), int(rng.integers(20, 160)))].tobytes())
        s = int(rng.integers(0, max(1, len(x) - 10)))
        y = bytearray(x[s:s + int(rng.integers(4, 100))])
        for _ in range(int(rng.integers(0, 4))):                     # run-length errors: repeat or drop a base
            p = int(rng.integers(0, len(y)))
            if rng.random() < 0.5: y.insert(p, y[p])
            elif len(y) > 2: del y[p]
        y = y[:128]
        if k % 17 == 0: x[len(x) // 2] = ord("N")
        if k % 19 == 0: x = bytearray(bytes(x).lower())
        pb2.add(bytes(x), bytes(y), [int(v) for v in rng.choice([7, 12, 20, 30], len(y))], -1 if not banded else int(rng.integers(2, 9)))
    _check_homopoly(oracle, pb2, GapParams(), _hop(rng))


def test_homopolymer_mode_with_default_hop_parameters_equals_the_pair_hmm_kernel():
    pb, _ = realign_synth.generate(200, seed=14)
    from varlociraptor_amd.realign import HopParams
    a = realign.prob_related(pb)
    b = realign.prob_related_homopolymer(pb, GapParams(), HopParams())
    both_inf = np.isneginf(a) & np.isneginf(b)
    assert np.all(np.where(both_inf, 0.0, np.abs(a - b)) <= 1e-12 * np.maximum(1.0, np.abs(a)))


def test_homopolymer_mode_rejects_parameters_that_are_not_log_probabilities():
    from varlociraptor_amd.realign import HopParams
    from varlociraptor_amd import engine
    pb = PairBatch()
    pb.add(b"ACGT", b"CG", [30, 30])
    with pytest.raises(engine.EngineError):
        realign.prob_related_homopolymer(pb, GapParams(), HopParams([0.5, -1, -1, -1]))
