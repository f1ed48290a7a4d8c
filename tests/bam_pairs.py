"""Test-side front end of SURVEY §8 f1 with REAL input: (read window, allele window) pairs cut from BAM records the way the
reference's realigner does, for the engine's edit-distance and pair-HMM kernels (varlociraptor_amd/realign.py).

Test infrastructure, not product code: a minimal BAM / FASTA reader (BGZF members are gzip members), the projection of reference
positions into reads, and the candidate regions of one read against one variant locus.  Restates

  rust-htslib `CigarStringView::read_pos(ref_pos, include_softclips, include_dels)`   (third-party; semantics as documented there:
        leading soft clips shift the alignment start when they are included, a position inside a deletion projects to the read
        position at which the deletion starts, hard clips and pads consume nothing)
  Realigner::candidate_region                 realignment/mod.rs:58-153
  Realigner::ref_window / max_window          realignment/mod.rs:149-158 (ref window = 1.5 x realignment window)
  SingleLocus::overlap (enclosing / some)     variants/types/mod.rs (used by Deletion::is_valid_evidence, types/deletion.rs:167-175)

`indel_pairs` classifies the candidate as utils/collect_variants.rs:274-300 does (deletion / insertion with a one-base anchor,
anything else of unequal lengths a replacement) and builds the alt allele window of that type; `single_end_pileup` turns the
per-read supports into the engine's pileup.

What is NOT restated (and the tests say so where it matters): fragments (read pairs are two single-end observations here, without
the insert-size support of deletion.rs:232-258), alternative variants at the locus, the read-inferred third allele
(mod.rs:311-349), prob_sample_alt (taken as certain).
"""
from __future__ import annotations

import gzip
import math
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

MAX_PATTERN_LEN = 128   # edit_distance.rs:145-147
SEQ_CODE = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


@dataclass
class BamRecord:
    qname: str
    flag: int
    ref_id: int
    pos: int          # 0-based leftmost
    mapq: int
    cigar: List[Tuple[str, int]]
    seq: bytes
    qual: bytes
    mate_ref_id: int
    mate_pos: int
    tlen: int

    @property
    def reverse(self) -> bool:
        return bool(self.flag & 0x10)

    @property
    def unmapped(self) -> bool:
        return bool(self.flag & 0x4)

    def end_pos(self) -> int:
        """exclusive reference end of the alignment (CigarStringView::end_pos)"""
        return self.pos + sum(l for op, l in self.cigar if op in "MDN=X")


def read_bam(path: str) -> Tuple[List[Tuple[str, int]], List[BamRecord]]:
    """(contigs, records) of a BAM file (SAM spec §4.2)."""
    d = gzip.open(path).read()
    assert d[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", d, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, o)
    o += 4
    contigs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", d, o)
        name = d[o + 4:o + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<i", d, o + 4 + l_name)
        contigs.append((name, l_ref))
        o += 8 + l_name
    recs = []
    while o < len(d):
        block_size, = struct.unpack_from("<i", d, o)
        ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, mate_ref, mate_pos, tlen = struct.unpack_from("<iiBBHHHiiii", d, o + 4)
        p = o + 36
        qname = d[p:p + l_read_name - 1].decode()
        p += l_read_name
        cigar = []
        for k in range(n_cigar):
            v, = struct.unpack_from("<I", d, p + 4 * k)
            cigar.append((CIGAR_OPS[v & 0xf], v >> 4))
        p += 4 * n_cigar
        packed = d[p:p + (l_seq + 1) // 2]
        seq = bytearray(l_seq)
        for i in range(l_seq):
            b = packed[i >> 1]
            seq[i] = ord(SEQ_CODE[(b >> 4) if (i & 1) == 0 else (b & 0xf)])
        p += (l_seq + 1) // 2
        qual = d[p:p + l_seq]
        recs.append(BamRecord(qname, flag, ref_id, pos, mapq, cigar, bytes(seq), bytes(qual), mate_ref, mate_pos, tlen))
        o += 4 + block_size
    return contigs, recs


def read_fasta(path: str) -> Dict[str, bytes]:
    out: Dict[str, bytearray] = {}
    name = None
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            out[name] = bytearray()
        elif name is not None:
            out[name] += line.encode()
    return {k: bytes(v) for k, v in out.items()}


def read_pos(rec: BamRecord, ref_pos: int, include_softclips: bool, include_dels: bool) -> Optional[int]:
    """rust-htslib CigarStringView::read_pos."""
    cig = rec.cigar
    rpos, qpos, j = rec.pos, 0, 0
    for i, (op, l) in enumerate(cig):
        if op in "M=XI":
            j = i
            break
        if op == "S":
            j = i
            if include_softclips:
                rpos = max(0, rpos - l)
            break
        if op == "D":
            rpos += l
        elif op == "N":
            raise ValueError("leading reference skip")
        elif op in "HP":
            if i == len(cig) - 1:
                return None
    while rpos <= ref_pos and j < len(cig):
        op, l = cig[j]
        inside = rpos <= ref_pos < rpos + l
        if op in "M=X" and inside:
            return qpos + (ref_pos - rpos)
        if op == "S" and include_softclips and inside:
            return qpos + (ref_pos - rpos)
        if op == "D" and include_dels and inside:
            return qpos
        if op in "M=X":
            rpos += l; qpos += l
        elif op == "S":
            qpos += l
            if include_softclips:
                rpos += l
        elif op == "I":
            qpos += l
        elif op in "DN":
            rpos += l
        elif op == "H" and j == len(cig) - 1:
            return None
        j += 1
    return None


def overlaps(rec: BamRecord, start: int, end: int) -> bool:
    """the alignment (with soft clips) shares at least one position with [start, end): SingleLocus::overlap(read, true, ..) != None"""
    lead = rec.cigar[0][1] if rec.cigar and rec.cigar[0][0] == "S" else 0
    trail = rec.cigar[-1][1] if len(rec.cigar) > 1 and rec.cigar[-1][0] == "S" else 0
    return rec.pos - lead < end and rec.end_pos() + trail > start


@dataclass
class CandidateRegion:
    overlap: bool
    read_interval: Tuple[int, int]
    ref_interval: Tuple[int, int]


def candidate_region(rec: BamRecord, locus_start: int, locus_end: int, ref_len: int, window: int = 64) -> CandidateRegion:
    """Realigner::candidate_region (realignment/mod.rs:58-153); `window` = --realignment-window (max_window), ref window = 1.5 x."""
    ref_window = int(window * 1.5)
    seq_len = len(rec.seq)

    def ref_interval(bp: int) -> Tuple[int, int]:
        return max(0, bp - ref_window), min(bp + ref_window, ref_len)

    qs, qe = read_pos(rec, locus_start, True, True), read_pos(rec, locus_end, True, True)
    if qs is not None and qe is not None:
        max_window = max(0, window - (qe - qs) // 2)
        ro, re_ = max(0, qs - max_window), min(qe + max_window, seq_len)
        exceed = max(0, (re_ - ro) - MAX_PATTERN_LEN)
        if exceed > 0:
            ro += exceed // 2
            re_ -= int(math.ceil(exceed / 2.0))
        return CandidateRegion(True, (ro, re_), ref_interval(locus_start))
    if qs is not None:
        return CandidateRegion(True, (max(0, qs - window), min(qs + window, seq_len)), ref_interval(locus_start))
    if qe is not None:
        return CandidateRegion(True, (max(0, qe - window), min(qe + window, seq_len)), ref_interval(locus_end))
    m = seq_len // 2
    enclosed = rec.pos >= locus_start and rec.end_pos() <= locus_end
    return CandidateRegion(enclosed, (max(0, m - window), min(m + window - 1, seq_len)), ref_interval(rec.pos + m))


def prob_mapping(mapq: int) -> float:
    """ln(1 - 10^(-MAPQ/10)) (read_observation.rs:620-640; MAPQ 0 -> ln 0)"""
    p = 1.0 - 10.0 ** (-mapq / 10.0)
    return math.log(p) if p > 0.0 else -math.inf


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
    from varlociraptor_amd import realign
    bam, = glob.glob(os.path.join(d, "*.bam"))
    _, recs = read_bam(bam)
    var = open(os.path.join(d, "variant.tsv")).read().split("\n")[1].split("\t")
    ref_seq = read_fasta(os.path.join(d, "ref.fa"))[var[0]].upper()
    start = int(var[1]) - 1
    ref, alt = var[3].encode(), var[4].encode()
    assert ref_seq[start:start + len(ref)] == ref and len(ref) != len(alt)
    ref_window = int(window * 1.5)
    n = len(ref_seq)
    if len(alt) == 1 and ref[:1] == alt:
        # Deletion::new: locus = start..end, deleted bases start+1..end (deletion.rs:41-49); alt window deletion.rs:117-137
        kind, dl = "deletion", len(ref) - 1
        end = start + dl
        allele = realign.deletion_allele(ref_seq, max(0, start - ref_window), min(start + ref_window, n - dl), start, dl)
    elif len(ref) == 1 and alt[:1] == ref:
        # Insertion::new: locus = start..start+1; alt window insertion.rs:92-113
        kind, ins = "insertion", alt[1:]
        end = start + 1
        allele = realign.insertion_allele(ref_seq, max(0, start - ref_window), min(start + len(ins) + ref_window, n), start, ins)
    else:
        # Replacement::new: locus = the REF allele's interval; alt window replacement.rs:73-103
        kind = "replacement"
        end = start + len(ref)
        allele = realign.replacement_allele(ref_seq, max(0, start - ref_window), min(start + len(ref) + ref_window, n), start, len(ref), alt)
    reads = []
    for r in recs:
        if r.unmapped or r.flag & 0x900 or not overlaps(r, start, end):   # (secondary / supplementary records carry no evidence)
            continue
        reg = candidate_region(r, start, end, n, window)
        if not reg.overlap:
            continue
        ro, re_ = reg.read_interval
        if re_ - ro < 1:
            continue
        reads.append((r, r.seq[ro:re_].upper(), list(r.qual[ro:re_]), realign.ref_allele(ref_seq, *reg.ref_interval)))
    return IndelCase(kind, start, end, len(alt) - len(ref), allele, reads)


def pair_batch(case: IndelCase):
    """(ref allele, read), (alt allele, read) per read, bands unset"""
    from varlociraptor_amd.realign import PairBatch
    pb = PairBatch()
    for r, seq, qual, ref_allele in case.reads:
        pb.add(ref_allele, seq, qual, -1)
        pb.add(case.alt_allele, seq, qual, -1)
    return pb


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
