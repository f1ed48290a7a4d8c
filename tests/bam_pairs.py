"""Test-side layer over varlociraptor_amd/readwindows.py (the product's BAM front end of SURVEY §8 f1): the reference testcases held
under tests/golden/bam/, how a testcase directory becomes (read window, allele window) pairs, and how per-read supports become the
engine's pileup.  What readwindows.py does not mirror — fragments (mates are two single-end observations, no insert-size support),
alternative variants at the locus, the read-inferred third allele, prob_sample_alt — is not restated here either; the tests say so where it
matters."""
from __future__ import annotations

import math
from dataclasses import dataclass

from varlociraptor_amd.readwindows import (MAX_PATTERN_LEN, BamRecord, CandidateRegion, IndelLocus, candidate_region, evidence_windows,  # noqa: F401
                                           indel_locus, overlaps, prob_mapping, read_bam, read_fasta, read_pos)
from varlociraptor_amd.readwindows import pair_batch as _pair_batch


@dataclass
class IndelCase:
    kind: str                 # "deletion" | "insertion" | "replacement"
    start: int                # locus start (0-based; the anchor base of a deletion / insertion)
    end: int                  # locus end (exclusive)
    len_diff: int             # alt length - ref length
    alt_allele: bytes         # the alt allele window (independent of the read)
    reads: list               # (record, read window, qualities, ref allele window) per read that is valid evidence


def indel_pairs(d: str, window: int = 64) -> IndelCase:
    """Reads of the testcase in directory `d` (one *.bam, ref.fa, variant.tsv) that are valid evidence for its variant, with their
    read windows and allele windows."""
    import glob
    import os
    bam, = glob.glob(os.path.join(d, "*.bam"))
    _, recs = read_bam(bam)
    var = open(os.path.join(d, "variant.tsv")).read().split("\n")[1].split("\t")
    ref_seq = read_fasta(os.path.join(d, "ref.fa"))[var[0]].upper()
    loc = indel_locus(ref_seq, int(var[1]) - 1, var[3].encode(), var[4].encode(), window)
    return IndelCase(loc.kind, loc.start, loc.end, loc.len_diff, loc.alt_allele, evidence_windows(recs, ref_seq, loc, window))


def pair_batch(case: IndelCase):
    """(ref allele, read), (alt allele, read) per read, bands unset"""
    return _pair_batch(case.reads, case.alt_allele)


def single_end_pileup(case: IndelCase, pa, pr):
    """One observation per read from its normalised supports (ln P(read | alt), ln P(read | ref)): certain sampling, no
    double-overlap term, uniform hit probability over the read, strand from the record — and a sample model without artifact
    hypotheses (locus_flags 0), since strand / orientation / position features of a FRAGMENT need the mate logic this front end
    does not have."""
    import numpy as np
    from varlociraptor_amd import abi
    from varlociraptor_amd.batch import PileupBatch
    recs = [r for r, _, _, _ in case.reads]
    n = len(recs)
    pa, pr = np.asarray(pa, float), np.asarray(pr, float)
    strand = np.where(pa != pr, np.where([r.reverse for r in recs], abi.STRAND_REVERSE, abi.STRAND_FORWARD), abi.STRAND_NONE)
    cols = {
        "prob_mapping": [prob_mapping(r.mapq) for r in recs], "prob_alt": pa, "prob_ref": pr,
        "prob_missed_allele": np.logaddexp(pa, pr) - math.log(2.0),                  # types/mod.rs:100-102
        "prob_sample_alt": np.zeros(n), "prob_double_overlap": np.full(n, -np.inf),
        "prob_hit_base": [-math.log(float(len(r.seq))) for r in recs],
        "flags": abi.pack_flags(strand, np.full(n, abi.ORIENT_NONE), np.zeros(n, bool), np.zeros(n, bool), np.ones(n, bool),
                                np.array([r.mapq == 60 for r in recs]), np.full(n, abi.ALTLOCUS_NONE)),
    }
    return PileupBatch(1, np.array([0, n], np.uint32), {k: np.asarray(v, np.float32) if k != "flags" else v for k, v in cols.items()},
                       {"locus_flags": np.array([0], np.uint8), "variant_type": np.array([abi.VT_INDEL], np.uint8)})


def phred_by_event(scenario, ln_posterior_row):
    """{"PROB_<EVENT>": PHRED of the event's posterior} as the reference's testcase runner reads them from the INFO column"""
    return {"PROB_" + name.upper(): -10.0 * float(ln_posterior_row[1 + k]) / math.log(10.0) for k, name in enumerate(scenario.event_names)}


# the reference testcases held under tests/golden/bam/ (copied by tools/make_bam_fixtures.py): scenario file, contig of the ploidy
# lookup, gap parameters of the sample (testcase.yaml: alignment properties where recorded, GapParams::default otherwise), the
# variant type utils/collect_variants.rs gives the candidate, and the `expected:` block as a predicate over (MAP allele frequency,
# PHRED posterior by event)
BAM_CASES = {
    "test_false_negative_indel_call": dict(        # tests/lib.rs:192; MN908947.3:517 TATG>T; `sample > 0.0`, `PROB_PRESENT <= 0.05`
        scenario=("testcases", "test_false_negative_indel_call", "scenario.yaml"), contig=None,
        gap=(-12.785891140783116, -12.186270018233994, -math.inf, -math.inf), kind="deletion", n_reads=342,
        expected=lambda vaf, ph: vaf > 0.0 and ph["PROB_PRESENT"] <= 0.05 and 0.02 < vaf < 0.6),
    "test_giab_04": dict(                           # lib.rs:105; 1:1201 GAAAAAAAAATACAG>GAAAAAAAATACAG; `NA12878 == 1.0`, `PROB_PRESENT <= 1.0`
        scenario=("bam", "test_giab_04", "scenario.yaml"), contig="1", gap=None, kind="replacement", n_reads=40,
        expected=lambda vaf, ph: vaf == 1.0 and ph["PROB_PRESENT"] <= 1.0),
    "test_giab_05": dict(                           # lib.rs:107; 1:1001 CCA>CGCC; `NA12878 == 1.0`
        scenario=("bam", "test_giab_05", "scenario.yaml"), contig="1", gap=None, kind="replacement", n_reads=129,
        expected=lambda vaf, ph: vaf == 1.0),
    "test_giab_06": dict(                           # lib.rs:109; 22:1200 G>GC; `index == 0.5`
        scenario=("bam", "test_giab_06", "scenario.yaml"), contig="22", gap=None, kind="insertion", n_reads=117,
        expected=lambda vaf, ph: vaf == 0.5),
    "test_giab_11": dict(                           # lib.rs:114; 1:1198 CCCCTCCCTCCCTCCCA>CCCCTCCCTCCCTCCCTCCCA; `PROB_HET <= 0.05 || PROB_HOM <= 0.05`
        scenario=("bam", "test_giab_11", "scenario.yaml"), contig="1", gap=None, kind="replacement", n_reads=232,
        expected=lambda vaf, ph: ph["PROB_HET"] <= 0.05 or ph["PROB_HOM"] <= 0.05),
    "test_giab_12": dict(                           # lib.rs:115; 1:1079 T>TCCT; `index == 0.5`
        scenario=("bam", "test_giab_12", "scenario.yaml"), contig="1", gap=None, kind="insertion", n_reads=149,
        expected=lambda vaf, ph: vaf == 0.5),
    "test_nanopore_05": dict(                       # lib.rs:169 testcase!(test_nanopore_05, homopolymer); chr1:111 T>TT in a T run, 26 reads of
        # ~224 bases; `PROB_GERMLINE_HET < 1.0 || PROB_GERMLINE_HOM < 1.0`; gap and homopolymer-run parameters of the sample's alignment properties
        scenario=("bam", "test_nanopore_05", "scenario.yaml"), contig="chr1", kind="insertion", n_reads=26,
        gap=(-4.320603998501797, -3.4073426177904595, -0.4111023053215838, -0.3729407271973288),
        hop=([-2.4241876197016907, -2.0631363999353076, -2.075720902621972, -2.4607220391071825],
             [-1.1525743935495203, -0.7538486971814485, -0.7904797474537245, -1.1450692553725095],
             [-4.74532040606337, -4.029387645256212, -3.963866371464708, -5.121149314030805],
             [-2.524995685227039, -3.137290211542217, -3.0427733560338037, -2.6016763482417575]),
        # in this mode the reads that spell the shorter run are explained by a homopolymer-run error of the read: the call is homozygous
        # (with the exact pair HMM on the same windows: heterozygous — tests/test_bam_pairs.py holds both)
        expected=lambda vaf, ph: (ph["PROB_GERMLINE_HET"] < 1.0 or ph["PROB_GERMLINE_HOM"] < 1.0) and vaf == 1.0),
    "test_giab_16": dict(                           # lib.rs:121; 1:1156 ATTTTTTTTTTATAGC>ATTTTTTTTTATAGC; `index != 0.0`
        scenario=("bam", "test_giab_16", "scenario.yaml"), contig="1", gap=None, kind="replacement", n_reads=172,
        expected=lambda vaf, ph: vaf != 0.0),
}
