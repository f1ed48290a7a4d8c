"""Read windows and allele windows from BAM records (the "next" row: the pair-HMM kernels fed with BAM-derived pairs).

Host side of `varlociraptor preprocess variants` for indel-like candidates, as far as the realignment needs it: which records are
evidence for a candidate, which window of a read is realigned, against which windows of the reference and of the alt allele, and the
per-read supports that come out of the edit-distance and pair-HMM kernels (`realign.best_hits`, `realign.prob_related`,
`realign.prob_related_homopolymer` — there is no CPU path).  Mirrors

  rust-htslib `CigarStringView::read_pos(ref_pos, include_softclips, include_dels)`   (third-party; semantics as documented there:
        leading soft clips shift the alignment start when they are included, a position inside a deletion projects to the read
        position at which the deletion starts, hard clips and pads consume nothing)
  Realigner::candidate_region                 realignment/mod.rs:58-153
  Realigner::ref_window / max_window          realignment/mod.rs:149-158 (ref window = 1.5 x realignment window)
  SingleLocus::overlap (enclosing / some)     variants/types/mod.rs (used by {Deletion,Insertion,Replacement}::is_valid_evidence)
  candidate classification                    utils/collect_variants.rs:274-300 (deletion / insertion with a one-base anchor, anything
                                              else of unequal lengths a replacement)
  alt allele windows                          types/deletion.rs:117-137, types/insertion.rs:92-113, types/replacement.rs:73-103
  Realigner::allele_support (single locus)    realignment/mod.rs:161-424: edit-distance hit -> banded pair HMM -> normalisation

NOT mirrored (callers must know): fragments — a read pair is two single-end observations here, without the insert-size support of
deletion.rs:232-258 —, alternative variants at the locus, the read-inferred third allele (mod.rs:311-349), `prob_sample_alt`,
SNVs / MNVs (scored base by base in the reference, no realignment).  A minimal BAM / FASTA reader is included (BGZF members are gzip
members): the reference reads through htslib.
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
class IndelLocus:
    kind: str                 # "deletion" | "insertion" | "replacement"
    start: int                # locus start (0-based; the anchor base of a deletion / insertion)
    end: int                  # locus end (exclusive)
    len_diff: int             # alt length - ref length
    alt_allele: bytes         # the alt allele window (independent of the read)


def indel_locus(ref_seq: bytes, pos: int, ref: bytes, alt: bytes, window: int = 64) -> IndelLocus:
    """Locus and alt allele window of the candidate `ref` > `alt` at 0-based `pos` of the contig sequence `ref_seq`."""
    from . import realign
    ref, alt = bytes(ref).upper(), bytes(alt).upper()
    if ref_seq[pos:pos + len(ref)].upper() != ref:
        raise ValueError("REF allele does not match the reference sequence at position %d" % (pos + 1))
    if len(ref) == len(alt):
        raise ValueError("SNVs and MNVs are not realigned (types/snv.rs, types/mnv.rs)")
    ref_window = int(window * 1.5)
    n = len(ref_seq)
    if len(alt) == 1 and ref[:1] == alt:
        # Deletion::new: locus = start..end, deleted bases start+1..end (deletion.rs:41-49); alt window deletion.rs:117-137
        dl = len(ref) - 1
        return IndelLocus("deletion", pos, pos + dl, -dl, realign.deletion_allele(ref_seq, max(0, pos - ref_window), min(pos + ref_window, n - dl), pos, dl))
    if len(ref) == 1 and alt[:1] == ref:
        # Insertion::new: locus = start..start+1; alt window insertion.rs:92-113
        ins = alt[1:]
        return IndelLocus("insertion", pos, pos + 1, len(ins), realign.insertion_allele(ref_seq, max(0, pos - ref_window), min(pos + len(ins) + ref_window, n), pos, ins))
    # Replacement::new: locus = the REF allele's interval; alt window replacement.rs:73-103
    return IndelLocus("replacement", pos, pos + len(ref), len(alt) - len(ref),
                      realign.replacement_allele(ref_seq, max(0, pos - ref_window), min(pos + len(ref) + ref_window, n), pos, len(ref), alt))


def evidence_windows(records: List[BamRecord], ref_seq: bytes, locus: IndelLocus, window: int = 64) -> list:
    """(record, read window, qualities, ref allele window) of every record that is valid evidence for the locus (mapped, primary,
    overlapping it with its soft clips, with a candidate region)."""
    from . import realign
    out = []
    n = len(ref_seq)
    for r in records:
        if r.unmapped or r.flag & 0x900 or not overlaps(r, locus.start, locus.end):   # (secondary / supplementary records carry no evidence)
            continue
        reg = candidate_region(r, locus.start, locus.end, n, window)
        if not reg.overlap:
            continue
        ro, re_ = reg.read_interval
        if re_ - ro < 1:
            continue
        out.append((r, r.seq[ro:re_].upper(), list(r.qual[ro:re_]), realign.ref_allele(ref_seq, *reg.ref_interval)))
    return out


def pair_batch(reads: list, alt_allele: bytes):
    """(ref allele, read), (alt allele, read) per read as a realign.PairBatch, bands unset"""
    from .realign import PairBatch
    pb = PairBatch()
    for r, seq, qual, ref_allele in reads:
        pb.add(ref_allele, seq, qual, -1)
        pb.add(alt_allele, seq, qual, -1)
    return pb


def allele_supports(reads: list, alt_allele: bytes, gap=None, hop=None, device: int = 0):
    """(ln P(read | alt), ln P(read | ref)) per read, normalised (Realigner::allele_support, mod.rs:161-424, one locus, no
    alternative variants): the edit-distance kernel bands every pair (best hit + EDIT_BAND, pairhmm.rs:20), the pair-HMM kernel scores
    it — `hop` (realign.HopParams) selects the homopolymer mode —, mod.rs:359-385 normalises.  Returns (prob_alt, prob_ref, PairBatch)."""
    import numpy as np
    from . import realign
    pb = pair_batch(reads, alt_allele)
    dist, _, _ = realign.best_hits(pb, device)
    pb.band = [int(x) + realign.EDIT_BAND if x >= 0 else -1 for x in dist]
    lnp = realign.prob_related_homopolymer(pb, gap, hop, device) if hop is not None else realign.prob_related(pb, gap, device)
    n = len(reads)
    pa, pr = np.empty(n), np.empty(n)
    for k in range(n):
        pr[k], pa[k] = realign.normalize_support(float(lnp[2 * k]), float(lnp[2 * k + 1]))
    return pa, pr, pb
