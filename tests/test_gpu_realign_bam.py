"""SURVEY §8 f1 with REAL input (VERDICT r05 missing #1): the read windows and allele windows of a reference testcase's own BAM go
through the engine's edit-distance and pair-HMM kernels, the supports become a pileup, and the call must satisfy the testcase's
own `expected:` block.

`test_false_negative_indel_call` (tests/lib.rs:192; testcase.yaml: `sample > 0.0`, `PROB_PRESENT <= 0.05` PHRED) is a 3-base deletion
(MN908947.3:517 TATG>T) every read of which goes through `Realigner::allele_support` (types/deletion.rs:201-213).  The observations
RECORDED in its candidates.vcf predate the fix the testcase documents (tests/test_oracle_fixture.py pins them: MAP VAF 0,
P(present) 0.465 — the reported false negative); the reference's test recomputes them from sample.bam with the fixed code.  So does
this test, with the engine's kernels in the place of `bio::stats::pairhmm`:

  BAM records -> candidate regions (tests/bam_pairs.py: realignment/mod.rs:58-153) -> (allele window, read window) pairs
  -> vlr_edit_distance_batch (band) + vlr_realign_batch (pair HMM, gap parameters of the testcase's alignment properties)
  -> normalisation (mod.rs:359-385) -> one observation per read -> vlr_batch_run -> MAP allele frequency and PROB_PRESENT.

What the pileup does not have (bam_pairs.py says so): fragments — mates are two observations instead of one merged support with
the insert-size term (deletion.rs:232-258) —, the read-inferred third allele, prob_sample_alt.  None of them can turn a carried
deletion into a reference read; the `expected:` block is an inequality for exactly that reason.  The same pairs are also compared
with the CPU restatement of the pair HMM (oracle/vlr_realign_oracle.cpp) to 1e-9 in ln P: parity on real windows, not only on
realign_synth's."""
import math
import os

import numpy as np
import pytest

import bam_pairs as bp
from varlociraptor_amd import abi, cli, engine, realign
from varlociraptor_amd.batch import PileupBatch
from varlociraptor_amd.realign import GapParams, PairBatch

pytestmark = pytest.mark.gpu
WINDOW = 64  # realignment_window of the testcase's recorded options


def _deletion_pairs(d):
    """reads of the testcase that are valid evidence for its deletion, with their read windows and allele windows"""
    contigs, recs = bp.read_bam(os.path.join(d, "sample.bam"))
    var = open(os.path.join(d, "variant.tsv")).read().split("\n")[1].split("\t")
    ref_seq = bp.read_fasta(os.path.join(d, "ref.fa"))[var[0]].upper()
    start = int(var[1]) - 1                      # the anchor base: position before the first deleted base
    del_len = len(var[3]) - len(var[4])
    end = start + del_len                        # Deletion::new: locus = start..end, deleted bases start+1..end (deletion.rs:41-49)
    assert ref_seq[start:start + len(var[3])] == var[3].encode()
    ref_window = int(WINDOW * 1.5)
    # alt allele window (Deletion::alt_emission_params, deletion.rs:117-137): independent of the read
    a_off, a_end = max(0, start - ref_window), min(start + ref_window, len(ref_seq) - del_len)
    alt_allele = realign.deletion_allele(ref_seq, a_off, a_end, start, del_len)
    reads = []
    for r in recs:
        if r.unmapped or r.flag & 0x900 or not bp.overlaps(r, start, end):   # (secondary / supplementary records carry no evidence)
            continue
        reg = bp.candidate_region(r, start, end, len(ref_seq), WINDOW)
        if not reg.overlap:
            continue
        ro, re_ = reg.read_interval
        if re_ - ro < 1:
            continue
        ref_allele = realign.ref_allele(ref_seq, *reg.ref_interval)
        reads.append((r, r.seq[ro:re_].upper(), list(r.qual[ro:re_]), ref_allele))
    return reads, alt_allele, start, del_len


def test_deletion_testcase_from_its_bam_meets_the_reference_expectation(oracle, golden_dir):
    name = "test_false_negative_indel_call"
    d = os.path.join(golden_dir, "bam", name)
    reads, alt_allele, start, del_len = _deletion_pairs(d)
    assert len(reads) > 300   # 342 records overlap the locus; the recorded pileup has 170 FRAGMENTS
    gap = GapParams(-12.785891140783116, -12.186270018233994, -math.inf, -math.inf)   # testcase.yaml: alignment properties of the sample
    # ---- pairs: (ref allele, read), (alt allele, read) per read; band = best hit's edit distance + EDIT_BAND (pairhmm.rs:20)
    pb = PairBatch()
    for r, seq, qual, ref_allele in reads:
        pb.add(ref_allele, seq, qual, -1)
        pb.add(alt_allele, seq, qual, -1)
    dist, _, _ = realign.best_hits(pb)
    host_dist = [realign.best_hit(pb.y[k], pb.x[k])[0] for k in range(0, len(pb), 37)]
    assert [int(dist[k]) for k in range(0, len(pb), 37)] == host_dist          # the edit-distance kernel on real windows
    pb.band = [int(x) + realign.EDIT_BAND if x >= 0 else -1 for x in dist]
    lnp = realign.prob_related(pb, gap)
    ref_lnp = oracle.pairhmm_batch(pb, gap, threads=8)
    both_inf = np.isneginf(lnp) & np.isneginf(ref_lnp)
    dev = np.where(both_inf, 0.0, np.abs(lnp - ref_lnp))
    assert np.all(dev <= 1e-9 * np.maximum(1.0, np.abs(ref_lnp) * 1e-3)), float(np.nanmax(dev))   # pair HMM: kernel == restatement
    # ---- one observation per read (single-end evidence)
    n = len(reads)
    pa, pr = np.empty(n), np.empty(n)
    for k in range(n):
        pr[k], pa[k] = realign.normalize_support(float(lnp[2 * k]), float(lnp[2 * k + 1]))
    carried = np.array([any(op == "D" and l == del_len for op, l in r.cigar) for r, _, _, _ in reads])
    # reads whose alignment carries the 3-base deletion support the alt allele, reads aligned through the locus without it the reference
    through = np.array([bp.read_pos(r, start, False, False) is not None and bp.read_pos(r, start + del_len + 1, False, False) is not None for r, _, _, _ in reads])
    assert (pa[carried] > pr[carried]).mean() > 0.95
    assert (pr[through & ~carried] > pa[through & ~carried]).mean() > 0.95
    pm = np.array([bp.prob_mapping(r.mapq) for r, _, _, _ in reads])
    miss = np.logaddexp(pa, pr) - math.log(2.0)                     # types/mod.rs:100-102
    strand = np.where(pa != pr, np.where([r.reverse for r, _, _, _ in reads], abi.STRAND_REVERSE, abi.STRAND_FORWARD), abi.STRAND_NONE)
    cols = {
        "prob_mapping": pm, "prob_alt": pa, "prob_ref": pr, "prob_missed_allele": miss, "prob_sample_alt": np.zeros(n),
        "prob_double_overlap": np.full(n, -np.inf), "prob_hit_base": np.full(n, -math.log(150.0)),
        "flags": abi.pack_flags(strand, np.full(n, abi.ORIENT_NONE), np.zeros(n, bool), np.zeros(n, bool), np.ones(n, bool),
                                np.array([r.mapq == 60 for r, _, _, _ in reads]), np.full(n, abi.ALTLOCUS_NONE)),
    }
    # the sample model without artifact hypotheses: strand / orientation / position features of a fragment need the mate logic this
    # front end does not have
    batch = PileupBatch(1, np.array([0, n], np.uint32), {k: np.asarray(v, np.float32) if k != "flags" else v for k, v in cols.items()},
                        {"locus_flags": np.array([0], np.uint8), "variant_type": np.array([abi.VT_INDEL], np.uint8)})
    sc = cli.scenario_from_yaml(os.path.join(golden_dir, "testcases", name, "scenario.yaml"))
    plan = engine.Plan(sc)
    got = plan.call_host(batch)
    plan.close()
    ref = oracle.call(sc, batch)
    assert np.allclose(np.exp(got.ln_posterior), np.exp(ref.ln_posterior), atol=1e-6, rtol=0) and abs(got.map_vaf[0, 0] - ref.map_vaf[0, 0]) <= 1e-6
    vaf = float(got.map_vaf[0, 0])
    phred_present = -10.0 * float(got.ln_posterior[0, 1]) / math.log(10.0)
    # the testcase's own expectation (testcase.yaml `expected:`; false on the observations recorded before the fix)
    assert vaf > 0.0, vaf
    assert phred_present <= 0.05, phred_present
    # and the deletion is carried by a minority of the reads: a frequency far from both ends
    assert 0.02 < vaf < 0.6, vaf
