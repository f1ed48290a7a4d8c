"""CPU checks of the test-side BAM front end (tests/bam_pairs.py) that feeds SURVEY §8 f1 with real (read window, allele window)
pairs: the reader on the reference's own BAM fixtures, the projection of reference positions into reads, the candidate regions of
realignment/mod.rs:58-153 — and, with the CPU restatement of the pair HMM in the kernels' place, the `expected:` blocks of
the three testcases held under tests/golden/bam/ (the GPU twin is tests/test_gpu_realign_bam.py)."""
import math
import os

import numpy as np
import pytest

import bam_pairs as bp
from bam_pairs import BamRecord


def _rec(pos, cigar, n=None):
    n = n if n is not None else sum(l for op, l in cigar if op in "MIS=X")
    return BamRecord("r", 0, 0, pos, 60, cigar, b"A" * n, bytes([30] * n), -1, -1, 0)


def test_read_pos_follows_the_cigar():
    r = _rec(100, [("S", 5), ("M", 10), ("D", 3), ("M", 10), ("I", 2), ("M", 5)])
    assert bp.read_pos(r, 100, False, False) == 5            # first aligned base
    assert bp.read_pos(r, 109, False, False) == 14
    assert bp.read_pos(r, 110, False, False) is None         # inside the deletion
    assert bp.read_pos(r, 110, False, True) == 15            # ... projects to where the deletion starts
    assert bp.read_pos(r, 113, False, False) == 15
    assert bp.read_pos(r, 123, False, False) == 27           # behind the insertion
    assert bp.read_pos(r, 97, False, False) is None and bp.read_pos(r, 97, True, False) == 2   # inside the leading soft clip
    assert bp.read_pos(r, 128, True, True) is None           # behind the read
    assert r.end_pos() == 128


def test_candidate_regions_of_enclosing_and_partial_reads():
    ref_len = 2000
    enclosing = _rec(400, [("M", 150)])
    reg = bp.candidate_region(enclosing, 516, 519, ref_len)
    assert reg.overlap and reg.ref_interval == (516 - 96, 516 + 96)
    qs, qe = 116, 119
    assert reg.read_interval == (qs - 63, min(qe + 63, 150))          # max_window shrinks by half the variant length
    left = _rec(380, [("M", 138)])                                      # ends inside the variant: only the start projects
    reg = bp.candidate_region(left, 516, 519, ref_len)
    assert reg.overlap and reg.read_interval == (136 - 64, 138) and reg.ref_interval == (420, 612)
    right = _rec(518, [("M", 150)])                                     # starts inside: only the end projects
    reg = bp.candidate_region(right, 516, 519, ref_len)
    assert reg.overlap and reg.read_interval == (0, 1 + 64) and reg.ref_interval == (519 - 96, 519 + 96)
    far = _rec(900, [("M", 150)])
    assert not bp.candidate_region(far, 516, 519, ref_len).overlap
    long_read = _rec(300, [("M", 400)])                                 # window capped at the longest pattern (128 bases)
    reg = bp.candidate_region(long_read, 516, 519, ref_len)
    assert reg.read_interval[1] - reg.read_interval[0] == bp.MAX_PATTERN_LEN


def test_reader_on_the_reference_fixtures(golden_dir):
    contigs, recs = bp.read_bam(os.path.join(golden_dir, "bam", "test_false_negative_indel_call", "sample.bam"))
    assert contigs == [("MN908947.3", 29903)] and len(recs) == 2870
    assert all(len(r.seq) == 150 == len(r.qual) == sum(l for op, l in r.cigar if op in "MIS=X") for r in recs)
    assert sum(1 for r in recs if not r.unmapped and bp.overlaps(r, 516, 519)) == 342
    fa = bp.read_fasta(os.path.join(golden_dir, "bam", "test_false_negative_indel_call", "ref.fa"))
    assert fa["MN908947.3"][516:520] == b"TATG"
    contigs, recs = bp.read_bam(os.path.join(golden_dir, "bam", "test_uzuner_clonal_1", "sample.bam"))
    assert len(recs) == 276 and contigs[5][0] == "chr6"


@pytest.mark.parametrize("name", sorted(bp.BAM_CASES))
def test_indel_testcase_from_its_bam_with_the_restated_pair_hmm(oracle, golden_dir, name):
    """The pipeline of tests/test_gpu_realign_bam.py with oracle/vlr_realign_oracle.cpp in the kernels' place: the pileup built from
    the testcase's BAM satisfies its `expected:` block — for `test_false_negative_indel_call` (`sample > 0.0`,
    `PROB_PRESENT <= 0.05`) the observations recorded before the fix do not (tests/test_oracle_fixture.py)."""
    from varlociraptor_amd import cli, realign
    from varlociraptor_amd.realign import GapParams
    spec = bp.BAM_CASES[name]
    case = bp.indel_pairs(os.path.join(golden_dir, "bam", name))
    assert case.kind == spec["kind"] and len(case.reads) == spec["n_reads"]
    gap = GapParams(*spec["gap"]) if spec["gap"] else GapParams()
    pb = bp.pair_batch(case)
    pb.band = [realign.best_hit(pb.y[k], pb.x[k])[0] + realign.EDIT_BAND for k in range(len(pb))]
    if spec.get("hop"):   # `homopolymer` mode (HomopolyPairHMM, realignment/mod.rs:680-730)
        from varlociraptor_amd.realign import HopParams
        lnp = oracle.homopoly_batch(pb, gap, HopParams(*spec["hop"]))
    else:
        lnp = oracle.pairhmm_batch(pb, gap, threads=8)
    n = len(case.reads)
    pa, pr = np.empty(n), np.empty(n)
    for k in range(n):
        pr[k], pa[k] = oracle.normalize_support(float(lnp[2 * k]), float(lnp[2 * k + 1]))
    if name == "test_false_negative_indel_call":
        carried = np.array([any(op == "D" and l == 3 for op, l in r.cigar) for r, _, _, _ in case.reads])
        assert carried.sum() == 40 and (pa[carried] > pr[carried]).all()
    sc = cli.scenario_from_yaml(os.path.join(golden_dir, *spec["scenario"]), **({"contig": spec["contig"]} if spec["contig"] else {}))
    res = oracle.call(sc, bp.single_end_pileup(case, pa, pr))
    assert (res.status[0] & 0xF) == 0
    assert spec["expected"](float(res.map_vaf[0, 0]), bp.phred_by_event(sc, res.ln_posterior[0]))
    if name == "test_nanopore_05":
        # the same windows through the exact pair HMM: the reads that spell the shorter run count against the insertion there
        lnp = oracle.pairhmm_batch(pb, gap, threads=8)
        for k in range(n):
            pr[k], pa[k] = oracle.normalize_support(float(lnp[2 * k]), float(lnp[2 * k + 1]))
        res = oracle.call(sc, bp.single_end_pileup(case, pa, pr))
        assert float(res.map_vaf[0, 0]) == 0.5


def test_product_front_end_classifies_candidates_and_refuses_what_is_not_realigned():
    """varlociraptor_amd/readwindows.py (the package's own BAM front end, which tests/bam_pairs.py wraps): candidate classes of
    utils/collect_variants.rs:274-300, loci of the three types, and SNVs / MNVs refused (they are scored base by base)."""
    from varlociraptor_amd import readwindows as rw
    ref = b"ACGTACGTTTTTACGATCGATCGGCTAGCTAGGATCGATTACA" * 8
    p = 100
    d = rw.indel_locus(ref, p, ref[p:p + 4], ref[p:p + 1])
    assert (d.kind, d.start, d.end, d.len_diff) == ("deletion", p, p + 3, -3) and len(d.alt_allele) == 192
    i = rw.indel_locus(ref, p, ref[p:p + 1], ref[p:p + 1] + b"GG")
    assert (i.kind, i.start, i.end, i.len_diff) == ("insertion", p, p + 1, 2) and len(i.alt_allele) == 196   # (insertion.rs:92-113: the window grows by the insertion on both counts)
    r = rw.indel_locus(ref, p, ref[p:p + 5], ref[p:p + 1] + b"TT" + ref[p + 4:p + 5])
    assert (r.kind, r.start, r.end, r.len_diff) == ("replacement", p, p + 5, -1)
    with pytest.raises(ValueError):
        rw.indel_locus(ref, p, ref[p:p + 1], b"N" if ref[p:p + 1] != b"N" else b"A")
    with pytest.raises(ValueError):
        rw.indel_locus(ref, p, b"NNNN", b"N")
    rec = _rec(p - 60, [("M", 150)])
    (got,) = rw.evidence_windows([rec], ref, d)
    assert got[0] is rec and len(got[1]) == len(got[2]) <= rw.MAX_PATTERN_LEN and len(got[3]) == 192
    assert rw.evidence_windows([_rec(p + 40, [("M", 50)])], ref, d) == []
