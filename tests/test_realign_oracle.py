"""CPU checks of the pair-HMM restatement (oracle/vlr_realign_oracle.cpp): closed forms for tiny cases, the emission and
gap conventions of realignment/pairhmm.rs, the normalisation of realignment/mod.rs:359-385, and the host-side
allele windows / edit-distance pre-filter of varlociraptor_amd/realign.py.  (The recursion itself lives in the bio crate,
absent from the reference tree: see the oracle's header — parity unpinned.)"""
import math

import numpy as np
import pytest

from varlociraptor_amd import realign

GAP = [math.log(2.8e-6), math.log(5.1e-6), -math.inf, -math.inf]


def test_single_cell_closed_form(oracle):
    """x = 'A', y = 'A' q30: paths = match from the start row: P = P(no gap) * (1 - 10^-3)."""
    p = oracle.pairhmm_prob_related(b"A", b"A", [30], GAP)
    want = math.log1p(-(2.8e-6 + 5.1e-6)) + math.log1p(-1e-3)
    assert abs(p - want) < 1e-12
    # mismatch: P(miscall) * 0.3333 (pairhmm.rs:22-24, 430-445)
    p = oracle.pairhmm_prob_related(b"C", b"A", [30], GAP)
    assert abs(p - (math.log1p(-(2.8e-6 + 5.1e-6)) + math.log(1e-3) + math.log(0.3333))) < 1e-12


def test_semiglobal_free_ends_sum_over_placements(oracle):
    """A read of one base against 'ACGT': one matching and three mismatching placements (gap paths add terms of order 1e-6);
    against 'AAAA' the four matching placements exceed one and the result is capped at ln 1."""
    pn = math.log1p(-(2.8e-6 + 5.1e-6))
    p4 = oracle.pairhmm_prob_related(b"ACGT", b"A", [40], GAP)
    want = pn + math.log((1 - 1e-4) + 3 * 1e-4 * 0.3333)
    assert abs(p4 - want) < 1e-5
    assert oracle.pairhmm_prob_related(b"AAAA", b"A", [40], GAP) == 0.0
    assert oracle.pairhmm_prob_related(b"A" * 50, b"A", [2], GAP) <= 0.0


def test_case_insensitive_and_band_wide_enough_changes_nothing(oracle):
    x, y, q = b"ACGTTGCAAGGCTTAACG", b"GTTGCTAGGC", [30] * 10
    a = oracle.pairhmm_prob_related(x, y, q, GAP)
    assert oracle.pairhmm_prob_related(x.lower(), y, q, GAP) == a
    # a band wider than the read admits every cell a path can reach; narrower bands only remove mass
    assert abs(oracle.pairhmm_prob_related(x, y, q, GAP, max_edit_dist=50) - a) < 1e-12
    assert oracle.pairhmm_prob_related(x, y, q, GAP, max_edit_dist=1) <= a + 1e-15
    hit = realign.best_hit(y, x)
    assert hit[0] == 1
    tight = oracle.pairhmm_prob_related(x, y, q, GAP, max_edit_dist=hit[0] + realign.EDIT_BAND)
    assert abs(tight - a) < 1e-9  # what the band drops is negligible


def test_gap_extension_parameters(oracle):
    """Without extension a two-base deletion needs mismatches; with it the gap path dominates."""
    x, y, q = b"ACGTACGTTTGGCCAATT", b"ACGTACGGGCCAATT"[:], [40] * 15
    no_ext = oracle.pairhmm_prob_related(x, y, q, GAP)
    ext = oracle.pairhmm_prob_related(x, y, q, [GAP[0], GAP[1], math.log(0.1), math.log(0.1)])
    assert ext > no_ext + 5.0


def test_normalize_support(oracle):
    r, a = oracle.normalize_support(math.log(1e-10), math.log(3e-10))
    assert abs(math.exp(r) + math.exp(a) - 1.0) < 1e-12 and abs(math.exp(a) - 0.75) < 1e-12
    assert oracle.normalize_support(-math.inf, -math.inf) == (math.log(0.5), math.log(0.5))
    assert oracle.normalize_support(-math.inf, -3.0) == (-math.inf, -3.0)  # no magnification against a zero (mod.rs:365-373)
    for pair in [(-5.0, -7.0), (-math.inf, -math.inf), (-math.inf, -1.0)]:
        assert realign.normalize_support(*pair) == pytest.approx(oracle.normalize_support(*pair), abs=1e-15)


def test_allele_windows_follow_the_emission_types():
    ref = b"AAAACCCCGGGGTTTTACGT"
    assert realign.ref_allele(ref, 2, 10) == b"AACCCCGG"
    assert realign.snv_allele(ref, 2, 10, 4, ord("T")) == b"AATCCCGG"
    assert realign.mnv_allele(ref, 2, 10, 4, b"TG") == b"AATGCCGG"
    # deletion of CCCC after position 3 (types/deletion.rs:316-323): window keeps its length, later bases shift in
    assert realign.deletion_allele(ref, 2, 10, 3, 4) == b"AAGGGGTT"
    # insertion of TT after position 3 (types/insertion.rs:252-274): window grows by the insertion
    assert realign.insertion_allele(ref, 2, 10, 3, b"TT") == b"AATTCCCCGG"
    # replacement of CCCC (4..8) by TG (types/replacement.rs:255-310): window shrinks by the length difference ...
    assert realign.replacement_allele(ref, 2, 12, 4, 4, b"TG") == b"AATGGGGG"
    # ... a replacement that shares its first base with the reference spells the deletion of the rest; one that is longer, the insertion
    assert realign.replacement_allele(ref, 2, 12, 3, 5, b"A") == realign.deletion_allele(ref, 2, 8, 3, 4) == b"AAGGGG"
    assert realign.replacement_allele(ref, 2, 10, 3, 1, b"ATT") == realign.insertion_allele(ref, 2, 10, 3, b"TT")


def test_best_hit_is_the_semiglobal_edit_distance():
    rng = np.random.default_rng(3)
    for _ in range(12):
        x = rng.choice(np.frombuffer(b"ACGT", np.uint8), 24).tobytes()
        y = rng.choice(np.frombuffer(b"ACGT", np.uint8), 8).tobytes()
        d, end = realign.best_hit(y, x)
        # brute force: minimum over substrings via plain DP
        best = min(_edit(y, x[s:e]) for s in range(len(x)) for e in range(s, min(len(x), s + 2 * len(y)) + 1))
        assert d == best
    assert realign.best_hit(b"ACGT", b"TTACGTTT") == (0, 6)


def _edit(a, b):
    D = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        prev, D[0] = D[0], i
        for j in range(1, len(b) + 1):
            cur = min(prev + (a[i - 1] != b[j - 1]), D[j] + 1, D[j - 1] + 1)
            prev, D[j] = D[j], cur
    return D[len(b)]


def test_reads_support_their_allele_of_origin(oracle):
    """End to end on the CPU oracle: reads drawn from the alt allele prefer it after normalisation, reference reads do not."""
    from varlociraptor_amd import realign_synth
    from varlociraptor_amd.realign import GapParams
    pb, truth = realign_synth.generate(40, seed=5, reads_per_locus=20)
    p = oracle.pairhmm_batch(pb, GapParams())
    agree = 0
    for k, from_alt in enumerate(truth):
        r, a = oracle.normalize_support(p[2 * k], p[2 * k + 1])
        agree += (a > r) == from_alt or abs(a - r) < 1e-3
    assert agree >= 38


def test_edit_distance_restatement_against_the_textbook_definition(oracle):
    """vlro_edit_distance = min over all substrings of the allele of the Levenshtein distance to the read (semiglobal),
    first end position, number of end positions: brute force on small random pairs, plus the numpy host routine."""
    from varlociraptor_amd import realign

    def lev(a, b):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
            prev = cur
        return prev[-1]

    rng = np.random.default_rng(17)
    B = np.frombuffer(b"ACGT", np.uint8)
    for _ in range(150):
        x = B[rng.integers(0, 4, int(rng.integers(1, 14)))].tobytes()
        y = B[rng.integers(0, 4, int(rng.integers(1, 9)))].tobytes()
        by_end = [min(lev(y, x[s:e]) for s in range(e + 1)) for e in range(1, len(x) + 1)]  # best start for every end position
        d = min(by_end)
        assert oracle.edit_distance(x, y) == (d, by_end.index(d) + 1, by_end.count(d)), (x, y)
        assert realign.best_hit(y, x) == (d, by_end.index(d) + 1)
    assert oracle.edit_distance(b"acgtACGT", b"CGTa")[0] == 0          # case-insensitive
    assert oracle.edit_distance(b"", b"A")[0] == -1 and oracle.edit_distance(b"A", b"")[0] == -1


def _brute_best_path(x, y, q, gap):
    """All semiglobal alignments of y in x by recursion: (edit distance, ln path probability) with the transition rules of
    PathHMMRealigner (realignment/mod.rs:598-660); returns the best ln p among those of minimal distance."""
    import math
    gx, gy, gxe, gye = gap
    lome = lambda p: math.log1p(-math.exp(p)) if p < -0.693 else (math.log(-math.expm1(p)) if p < 0 else -math.inf)
    lae = lambda a, b: max(a, b) + math.log1p(math.exp(-abs(a - b))) if max(a, b) > -math.inf else -math.inf
    no_gap = lome(lae(gx, gy))
    close_x, close_y = lome(gxe), lome(gye)
    reopen_x, reopen_y = lae(gxe, close_x + gx), lae(gye, close_y + gy)
    best = [None]

    def rec(i, j, prev, d, p):
        if j == len(y):
            if prev != "D" and (best[0] is None or (d, -p) < (best[0][0], -best[0][1])):
                best[0] = (d, p)
            return
        lm = -q[j] * math.log(10) / 10
        if i < len(x):  # match / substitution
            t = {None: 0.0, "M": no_gap, "D": close_y, "I": close_x}[prev]
            mm = x[i:i + 1].upper() != y[j:j + 1].upper()
            rec(i + 1, j + 1, "M", d + mm, p + t + (lm + math.log(0.3333) if mm else lome(lm)))
            if prev is not None:  # a leading deletion is never part of an optimal semiglobal alignment
                t = {"M": gy, "D": reopen_y, "I": close_x + gy}[prev]
                rec(i + 1, j, "D", d + 1, p + t)
        t = {None: gx, "M": gx, "I": reopen_x, "D": close_y + gx}[prev]
        rec(i, j + 1, "I", d + 1, p + t + lm)
    for start in range(len(x) + 1):
        rec(start, 0, None, 0, 0.0)
    return best[0]


def test_fast_mode_restatement_against_brute_force(oracle):
    """vlro_pathhmm_best (PathHMMRealigner, realignment/mod.rs:547-678) on tiny sequences: equal to the enumeration of all
    alignments — best path probability among those of minimal edit distance — with and without gap extension."""
    import math
    rng = np.random.default_rng(11)
    for gap in ([math.log(2.8e-6), math.log(5.1e-6), -math.inf, -math.inf], [math.log(1e-3), math.log(2e-3), math.log(0.2), math.log(0.3)]):
        for _ in range(60):
            lx, ly = int(rng.integers(1, 7)), int(rng.integers(1, 5))
            x = bytes(rng.choice(list(b"ACGT"), lx).tolist())
            y = bytes(rng.choice(list(b"ACGT"), ly).tolist())
            q = [int(v) for v in rng.choice([10, 20, 30, 40], ly)]
            d, p = _brute_best_path(x, y, q, gap)
            got = oracle.pathhmm_best(x, y, q, gap)
            assert got == pytest.approx(p, abs=1e-9), (x, y, q, d, p, got)
            assert oracle.edit_distance(x, y)[0] == d


# ---- homopolymer mode (HomopolyPairHMMRealigner, realignment/mod.rs:680-730; HopParams pairhmm.rs:207-295) -----------------

HOP0 = [-math.inf] * 16


def _brute_homopoly(x, y, q, gap, hop):
    """Sum over ALL state paths of the fourteen-state model as oracle/vlr_realign_oracle.cpp documents it (linear domain, plain
    recursion, no band): free start and end in x, Match_b labelled by the base of x_i, hops entered from / left to match states."""
    ex = lambda v: math.exp(v) if v > -math.inf else 0.0
    gx, gy, gxe, gye = [ex(v) for v in gap]
    idx = lambda c: b"ACGT".find(bytes([c]).upper())
    hx = [ex(v) for v in hop[0:4]]; hy = [ex(v) for v in hop[4:8]]; hxe = [ex(v) for v in hop[8:12]]; hye = [ex(v) for v in hop[12:16]]
    mis = [10 ** (-v / 10) for v in q]
    total = [0.0]

    def rec(i, j, st, b, p):
        if p == 0.0:
            return
        if j == len(y) and st != "S":
            total[0] += p                      # free end gap in x: every column may end the alignment, in any state
        # leave-to-match factor of the current state
        if st == "S": tm = 1 - (gx + gy)
        elif st == "M": tm = 1 - (gx + gy + (hx[b] + hy[b] if b >= 0 else 0.0))
        elif st == "GX": tm = 1 - gxe
        elif st == "GY": tm = 1 - gye
        elif st == "HX": tm = 1 - hxe[b]
        else: tm = 1 - hye[b]
        if i < len(x) and j < len(y):
            e = (1 - mis[j]) if bytes([x[i]]).upper() == bytes([y[j]]).upper() else mis[j] * 0.3333
            rec(i + 1, j + 1, "M", idx(x[i]), p * tm * e)
        if st in ("M", "GX") and j < len(y):   # y_j alone after a gap open / extension
            rec(i, j + 1, "GX", -1, p * (gx if st == "M" else gxe) * mis[j])
        if st in ("M", "GY") and i < len(x):   # x_i alone (prob_emit_x = 1)
            rec(i + 1, j, "GY", -1, p * (gy if st == "M" else gye))
        if st in ("M", "HX") and b >= 0 and j < len(y) and idx(y[j]) == b:
            rec(i, j + 1, "HX", b, p * (hx[b] if st == "M" else hxe[b]) * (1 - mis[j]))   # the repeated base is emitted like a matching base
        if st in ("M", "HY") and b >= 0 and i < len(x) and idx(x[i]) == b:
            rec(i + 1, j, "HY", b, p * (hy[b] if st == "M" else hye[b]))
    for start in range(len(x)):
        rec(start, 0, "S", -1, 1.0)
    return math.log(min(total[0], 1.0)) if total[0] > 0 else -math.inf


def test_homopolymer_mode_with_default_hop_parameters_is_the_pair_hmm(oracle):
    """HopParams::default() is all zero (pairhmm.rs:223-256): no hop state can be entered and the model is PairHMM."""
    rng = np.random.default_rng(21)
    for gap in (GAP, [math.log(1e-3), math.log(2e-3), math.log(0.2), math.log(0.3)]):
        for _ in range(40):
            x = bytes(rng.choice(list(b"ACGTN"), int(rng.integers(1, 40))).tolist())
            y = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 20))).tolist())
            q = [int(v) for v in rng.choice([10, 20, 30, 40], len(y))]
            for band in (-1, 3):
                a = oracle.pairhmm_prob_related(x, y, q, gap, band)
                b = oracle.homopoly_prob_related(x, y, q, gap, HOP0, band)
                assert (a == b) or abs(a - b) < 1e-12, (x, y, a, b)


def test_homopolymer_restatement_against_enumeration_of_all_paths(oracle):
    rng = np.random.default_rng(22)
    gaps = ([math.log(1e-3), math.log(2e-3), math.log(0.2), math.log(0.3)], GAP)
    for it in range(80):
        gap = gaps[it % 2]
        hop = [math.log(v) for v in rng.uniform(0.001, 0.05, 8)] + [math.log(v) for v in rng.uniform(0.05, 0.5, 8)]
        lx, ly = int(rng.integers(1, 7)), int(rng.integers(1, 5))
        alphabet = list(b"AAC") if it % 3 else list(b"ACGT")   # homopolymer-rich
        x = bytes(rng.choice(alphabet, lx).tolist())
        y = bytes(rng.choice(alphabet, ly).tolist())
        q = [int(v) for v in rng.choice([5, 10, 20, 30], ly)]
        want = _brute_homopoly(x, y, q, gap, hop)
        got = oracle.homopoly_prob_related(x, y, q, gap, hop)
        assert got == pytest.approx(want, abs=1e-10), (x, y, q, want, got)


def test_hop_parameters_explain_a_homopolymer_length_error(oracle):
    """A read with one T more than the allele's T run: with seq-homopolymer probabilities the insertion is explained by a hop,
    and only by a hop of the RIGHT base."""
    x, y, q = b"ACGTTTTGCA", b"CGTTTTTGC", [30] * 9
    plain = oracle.homopoly_prob_related(x, y, q, GAP, HOP0)
    hop_t = list(HOP0); hop_t[3] = math.log(0.05)            # prob_seq_homopolymer[T]
    hop_a = list(HOP0); hop_a[0] = math.log(0.05)            # ... [A]: does not apply to a T run
    assert oracle.homopoly_prob_related(x, y, q, GAP, hop_t) > plain + 5.0
    assert abs(oracle.homopoly_prob_related(x, y, q, GAP, hop_a) - plain) < 1e-3
    # one T less in the read: the reference-homopolymer (hop_y) parameters
    y2 = b"CGTTTGC"
    hop_ref = list(HOP0); hop_ref[4 + 3] = math.log(0.05)
    assert oracle.homopoly_prob_related(x, y2, [30] * 7, GAP, hop_ref) > oracle.homopoly_prob_related(x, y2, [30] * 7, GAP, HOP0) + 5.0
