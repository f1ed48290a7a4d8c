"""Host side of the read-vs-allele realignment path, mirroring the reference's interface:

  GapParams                 realignment/pairhmm.rs:119-143 (defaults: insertion 2.8e-6, deletion 5.1e-6, no extension)
  allele windows            ReferenceEmissionParams (pairhmm.rs:457-513), SNV / MNV / insertion / deletion emission params
                            (types/{snv,mnv,insertion,deletion}.rs `ref_base`): the allele sequence the read is compared with
  best_hit                  EditDistanceCalculation::calc_best_hit (edit_distance.rs:166-355): smallest semiglobal edit
                            distance of the read window against the allele window; the pair HMM is banded to dist + EDIT_BAND
  prob_related              PairHMMRealigner::calculate_prob_allele -> bio PairHMM::prob_related (mod.rs:519-537): the batched
                            HIP kernel behind vlr_realign_batch (include/vlr.h) — there is no CPU path here
  normalize_support         the ref/alt normalisation of Realigner::allele_support (mod.rs:359-385)

In reference v8.9.3 `shrink_to_hit` (pairhmm.rs:66-72) only moves the `ref_offset()/ref_end()` accessors; `ref_base` and
`len_x` of every emission type use the unshrunken fields, so the HMM always sees the whole reference window
(2 x 1.5 x realignment_window around the breakpoint, mod.rs:149-153) — the band is what bounds the work.
candidate_region / CIGAR projection and the per-read supports from BAM records: varlociraptor_amd/readwindows.py.
Not mirrored (documented in DESIGN.md): multiple loci per variant, the read-inferred "third allele" (mod.rs:311-349).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import abi, engine

EDIT_BAND = 4            # pairhmm.rs:20
MAX_PATTERN_LEN = 128    # edit_distance.rs:145-147


@dataclass
class GapParams:
    """ln probabilities (pairhmm.rs:119-143)."""
    prob_insertion_artifact: float = math.log(2.8e-6)
    prob_deletion_artifact: float = math.log(5.1e-6)
    prob_insertion_extend_artifact: float = -math.inf
    prob_deletion_extend_artifact: float = -math.inf

    def as_array(self):
        return (C.c_double * 4)(self.prob_insertion_artifact, self.prob_deletion_artifact,
                                self.prob_insertion_extend_artifact, self.prob_deletion_extend_artifact)


@dataclass
class HopParams:
    """ln probabilities per base A, C, G, T (pairhmm.rs:207-256; all zero probability by default)."""
    prob_seq_homopolymer: Sequence[float] = (-math.inf,) * 4
    prob_ref_homopolymer: Sequence[float] = (-math.inf,) * 4
    prob_seq_extend_homopolymer: Sequence[float] = (-math.inf,) * 4
    prob_ref_extend_homopolymer: Sequence[float] = (-math.inf,) * 4

    def as_list(self):
        v = [*self.prob_seq_homopolymer, *self.prob_ref_homopolymer, *self.prob_seq_extend_homopolymer, *self.prob_ref_extend_homopolymer]
        assert len(v) == 16
        return [float(t) for t in v]

    def as_array(self):
        return (C.c_double * 16)(*self.as_list())


class RealignDesc(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("x_offset", C.c_void_p), ("x_bases", C.c_void_p), ("y_offset", C.c_void_p),
                ("y_bases", C.c_void_p), ("y_quals", C.c_void_p), ("max_edit_dist", C.c_void_p), ("gap", C.c_double * 4)]


class PairBatch:
    """(allele window, read window) pairs in the layout of vlr_realign_batch_desc."""

    def __init__(self):
        self.x: List[bytes] = []
        self.y: List[bytes] = []
        self.q: List[bytes] = []
        self.band: List[int] = []

    def add(self, allele: bytes, read: bytes, qual: Sequence[int], max_edit_dist: int = -1):
        assert len(read) == len(qual)
        self.x.append(bytes(allele)); self.y.append(bytes(read)); self.q.append(bytes(bytearray(qual))); self.band.append(int(max_edit_dist))

    def __len__(self):
        return len(self.x)

    def arrays(self):
        xo = np.zeros(len(self) + 1, np.uint32); yo = np.zeros(len(self) + 1, np.uint32)
        xo[1:] = np.cumsum([len(v) for v in self.x]); yo[1:] = np.cumsum([len(v) for v in self.y])
        xb = np.frombuffer(b"".join(self.x) or b"\0", np.uint8).copy()
        yb = np.frombuffer(b"".join(self.y) or b"\0", np.uint8).copy()
        qb = np.frombuffer(b"".join(self.q) or b"\0", np.uint8).copy()
        return xo, xb, yo, yb, qb, np.asarray(self.band, np.int32)

    def cells(self) -> int:
        return int(sum(len(a) * len(b) for a, b in zip(self.x, self.y)))


def _bind():
    L = engine.lib()
    L.vlr_realign_batch_host.restype = C.c_int
    L.vlr_realign_batch_host.argtypes = [C.c_int, C.POINTER(RealignDesc), C.c_void_p]
    L.vlr_realign_batch.restype = C.c_int
    L.vlr_realign_batch.argtypes = [C.c_int, C.POINTER(RealignDesc), C.c_void_p, C.c_void_p]
    return L


def prob_related(batch: PairBatch, gap: Optional[GapParams] = None, device: int = 0) -> np.ndarray:
    """ln P(read window | allele) of every pair (vlr_realign_batch_host)."""
    L = _bind()
    gap = gap or GapParams()
    xo, xb, yo, yb, qb, band = batch.arrays()
    out = np.empty(len(batch), np.float64)
    d = RealignDesc(len(batch), xo.ctypes.data, xb.ctypes.data, yo.ctypes.data, yb.ctypes.data, qb.ctypes.data, band.ctypes.data, gap.as_array())
    rc = L.vlr_realign_batch_host(device, C.byref(d), out.ctypes.data)
    if rc != 0:
        raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
    return out


def prob_best_path(batch: PairBatch, gap: Optional[GapParams] = None, device: int = 0) -> np.ndarray:
    """`fast` mode (PathHMMRealigner, realignment/mod.rs:547-678): ln of the best path probability over the minimal-edit-distance
    alignments of every pair (vlr_realign_fast_batch_host)."""
    L = _bind()
    L.vlr_realign_fast_batch_host.restype = C.c_int
    L.vlr_realign_fast_batch_host.argtypes = [C.c_int, C.POINTER(RealignDesc), C.c_void_p]
    gap = gap or GapParams()
    xo, xb, yo, yb, qb, band = batch.arrays()
    out = np.empty(len(batch), np.float64)
    d = RealignDesc(len(batch), xo.ctypes.data, xb.ctypes.data, yo.ctypes.data, yb.ctypes.data, qb.ctypes.data, None, gap.as_array())
    rc = L.vlr_realign_fast_batch_host(device, C.byref(d), out.ctypes.data)
    if rc != 0:
        raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
    return out


def prob_related_homopolymer(batch: PairBatch, gap: Optional[GapParams] = None, hop: Optional[HopParams] = None, device: int = 0) -> np.ndarray:
    """`homopolymer` mode (HomopolyPairHMMRealigner, realignment/mod.rs:680-730): vlr_realign_homopolymer_batch_host."""
    L = _bind()
    L.vlr_realign_homopolymer_batch_host.restype = C.c_int
    L.vlr_realign_homopolymer_batch_host.argtypes = [C.c_int, C.POINTER(RealignDesc), C.POINTER(C.c_double), C.c_void_p]
    gap = gap or GapParams()
    hop = hop or HopParams()
    xo, xb, yo, yb, qb, band = batch.arrays()
    out = np.empty(len(batch), np.float64)
    d = RealignDesc(len(batch), xo.ctypes.data, xb.ctypes.data, yo.ctypes.data, yb.ctypes.data, qb.ctypes.data, band.ctypes.data, gap.as_array())
    rc = L.vlr_realign_homopolymer_batch_host(device, C.byref(d), hop.as_array(), out.ctypes.data)
    if rc != 0:
        raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
    return out


class DevicePairs:
    """A PairBatch resident in HBM (torch owns the buffers) for vlr_realign_batch / vlr_edit_distance_batch."""

    def __init__(self, batch: PairBatch, device="cuda:0"):
        import torch
        xo, xb, yo, yb, qb, band = batch.arrays()
        self.n = len(batch)
        self.t = [torch.from_numpy(a).to(device) for a in (xo.view(np.int32), xb, yo.view(np.int32), yb, qb, band)]
        self.out = torch.empty(self.n, dtype=torch.float64, device=device)
        self.cells = batch.cells()
        self.bytes = int(xb.nbytes + yb.nbytes + qb.nbytes + xo.nbytes + yo.nbytes + band.nbytes + self.n * 8)

    def run(self, gap: Optional[GapParams] = None, device: int = 0, stream: int = 0):
        L = _bind()
        gap = gap or GapParams()
        p = [t.data_ptr() for t in self.t]
        d = RealignDesc(self.n, p[0], p[1], p[2], p[3], p[4], p[5], gap.as_array())
        rc = L.vlr_realign_batch(device, C.byref(d), self.out.data_ptr(), stream)
        if rc != 0:
            raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
        return self.out

    def run_homopolymer(self, gap: Optional[GapParams] = None, hop: Optional[HopParams] = None, device: int = 0, stream: int = 0):
        L = _bind()
        L.vlr_realign_homopolymer_batch.restype = C.c_int
        L.vlr_realign_homopolymer_batch.argtypes = [C.c_int, C.POINTER(RealignDesc), C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
        gap = gap or GapParams()
        hop = hop or HopParams()
        p = [t.data_ptr() for t in self.t]
        d = RealignDesc(self.n, p[0], p[1], p[2], p[3], p[4], p[5], gap.as_array())
        rc = L.vlr_realign_homopolymer_batch(device, C.byref(d), hop.as_array(), self.out.data_ptr(), stream)
        if rc != 0:
            raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
        return self.out

    def band_from_hits(self, device: int = 0, stream: int = 0):
        """Edit-distance pre-filter on the resident pairs; sets the band of the pair HMM to distance + EDIT_BAND in place."""
        import torch
        L = _bind()
        L.vlr_edit_distance_batch.restype = C.c_int
        L.vlr_edit_distance_batch.argtypes = [C.c_int, C.POINTER(RealignDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        p = [t.data_ptr() for t in self.t]
        d = RealignDesc(self.n, p[0], p[1], p[2], p[3], p[4], None, GapParams().as_array())
        dist = torch.empty(self.n, dtype=torch.int32, device=self.t[5].device)
        rc = L.vlr_edit_distance_batch(device, C.byref(d), dist.data_ptr(), None, None, stream)
        if rc != 0:
            raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
        self.t[5].copy_(torch.where(dist >= 0, dist + EDIT_BAND, torch.full_like(dist, -1)))
        return dist


# ---- allele windows (`ref_base(i)` of the emission parameter types) -----------------------------------------------

def ref_allele(ref_seq: bytes, ref_offset: int, ref_end: int) -> bytes:
    """ReferenceEmissionParams (pairhmm.rs:457-513)."""
    return ref_seq[ref_offset:ref_end]


def snv_allele(ref_seq: bytes, ref_offset: int, ref_end: int, pos: int, alt: int) -> bytes:
    """types/snv.rs SNVEmissionParams::ref_base: the alt base at the locus, the reference elsewhere."""
    a = bytearray(ref_seq[ref_offset:ref_end])
    if ref_offset <= pos < ref_end:
        a[pos - ref_offset] = alt
    return bytes(a)


def mnv_allele(ref_seq: bytes, ref_offset: int, ref_end: int, start: int, alt: bytes) -> bytes:
    """types/mnv.rs:325-333."""
    a = bytearray(ref_seq[ref_offset:ref_end])
    for k, b in enumerate(alt):
        if ref_offset <= start + k < ref_end:
            a[start + k - ref_offset] = b
    return bytes(a)


def deletion_allele(ref_seq: bytes, ref_offset: int, ref_end: int, del_start: int, del_len: int) -> bytes:
    """types/deletion.rs:316-336: len_x = ref_end - ref_offset; i_ <= del_start reads the reference, later positions skip the
    deleted bases (del_start = the position before the first deleted base, as in the VCF)."""
    out = bytearray()
    for i in range(ref_end - ref_offset):
        i_ = i + ref_offset
        j = i_ if i_ <= del_start else i_ + del_len
        out.append(ref_seq[j] if j < len(ref_seq) else ord("N"))
    return bytes(out)


def insertion_allele(ref_seq: bytes, ref_offset: int, ref_end: int, ins_start: int, ins_seq: bytes) -> bytes:
    """types/insertion.rs:252-274: len_x = ref_end - ref_offset + ins_len; the inserted bases follow position ins_start."""
    ins_len = len(ins_seq)
    ins_end = ins_start + ins_len
    out = bytearray()
    for i in range(ref_end - ref_offset + ins_len):
        i_ = i + ref_offset
        if i_ <= ins_start:
            out.append(ref_seq[i_])
        elif i_ > ins_end:
            out.append(ref_seq[i_ - ins_len])
        else:
            out.append(ins_seq[i_ - (ins_start + 1)])
    return bytes(out)


def replacement_allele(ref_seq: bytes, ref_offset: int, ref_end: int, repl_start: int, repl_ref_len: int, repl_seq: bytes) -> bytes:
    """types/replacement.rs:255-310: the replacement sequence stands where the `repl_ref_len` reference bases from `repl_start` stood;
    len_x = ref_end - ref_offset + alt_len - ref_len (the unaltered length when that comes out as zero)."""
    alt_len = len(repl_seq)
    alt_end = repl_start + alt_len
    n = max(0, ref_end - ref_offset + alt_len - repl_ref_len) or (ref_end - ref_offset)
    out = bytearray()
    for i in range(n):
        i_ = i + ref_offset
        if i_ < repl_start:
            out.append(ref_seq[i_])
        elif i_ >= alt_end:
            j = i_ - alt_len + repl_ref_len
            out.append(ref_seq[j] if j < len(ref_seq) else ord("N"))
        else:
            out.append(repl_seq[i_ - repl_start])
    return bytes(out)


# ---- edit-distance pre-filter --------------------------------------------------------------------------------------

def best_hit(read: bytes, allele: bytes) -> Optional[Tuple[int, int]]:
    """Smallest edit distance of `read` against any substring of `allele` and the first end position reaching it
    (what bio's Myers `find_all_lazy` reports, edit_distance.rs:175-206).  Returns (dist, end) or None for empty input."""
    m, n = len(read), len(allele)
    if m == 0 or n == 0:
        return None
    x = np.frombuffer(bytes(allele).upper(), np.uint8)
    y = np.frombuffer(bytes(read).upper(), np.uint8)
    prev = np.zeros(n + 1, np.int64)  # free start in the allele
    ar = np.arange(n + 1, dtype=np.int64)
    for i in range(m):
        cur = np.empty(n + 1, np.int64)
        cur[0] = i + 1
        cur[1:] = np.minimum(prev[:-1] + (x != y[i]), prev[1:] + 1)
        cur = np.minimum.accumulate(cur - ar) + ar  # cur[j] = min(cur[j], cur[j-1] + 1)
        prev = cur
    d = int(prev[1:].min())
    return d, int(np.argmax(prev[1:] == d)) + 1


def best_hits(batch: "PairBatch", device: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(dist, end, n_hits) of every pair on the GPU (vlr_edit_distance_batch_host, csrc/vlr_realign.hip): what `best_hit` computes
    for one pair on the host, plus the number of end positions reaching the smallest distance."""
    L = _bind()
    L.vlr_edit_distance_batch_host.restype = C.c_int
    L.vlr_edit_distance_batch_host.argtypes = [C.c_int, C.POINTER(RealignDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    n = len(batch)
    dist = np.full(n, -1, np.int32)
    end = np.full(n, -1, np.int32)
    hits = np.zeros(n, np.int32)
    if n == 0:
        return dist, end, hits
    xo, xb, yo, yb, qb, band = batch.arrays()
    desc = RealignDesc(n, xo.ctypes.data, xb.ctypes.data, yo.ctypes.data, yb.ctypes.data, qb.ctypes.data, None, GapParams().as_array())
    rc = L.vlr_edit_distance_batch_host(device, C.byref(desc), dist.ctypes.data, end.ctypes.data, hits.ctypes.data)
    if rc != 0:
        raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
    return dist, end, hits


def normalize_support(prob_ref: float, prob_alt: float) -> Tuple[float, float]:
    """mod.rs:359-385."""
    if prob_ref != -math.inf and prob_alt != -math.inf:
        hi, lo = max(prob_ref, prob_alt), min(prob_ref, prob_alt)
        t = hi + math.log1p(math.exp(lo - hi))
        prob_ref, prob_alt = prob_ref - t, prob_alt - t
    if prob_ref == -math.inf and prob_alt == -math.inf:
        prob_ref = prob_alt = math.log(0.5)
    return prob_ref, prob_alt


def allele_support(reads: Sequence[Tuple[bytes, Sequence[int]]], ref_allele_seq: bytes, alt_allele_seq: bytes,
                   gap: Optional[GapParams] = None, device: int = 0) -> np.ndarray:
    """(prob_ref, prob_alt) of read windows against one reference and one alt allele window: edit-distance pre-filter,
    banded pair HMM on the GPU, normalisation (Realigner::allele_support for a single locus without alternative
    variants and without the read-inferred third allele)."""
    pb = PairBatch()
    for seq, qual in reads:
        assert len(seq) <= MAX_PATTERN_LEN
        for allele in (ref_allele_seq, alt_allele_seq):
            pb.add(allele, seq, qual, -1)
    dist, _, _ = best_hits(pb, device)  # edit-distance pre-filter on the GPU; the pair HMM is banded to distance + EDIT_BAND
    pb.band = [int(d) + EDIT_BAND if d >= 0 else -1 for d in dist]
    p = prob_related(pb, gap, device)
    out = np.empty((len(reads), 2))
    for k in range(len(reads)):
        out[k] = normalize_support(float(p[2 * k]), float(p[2 * k + 1]))
    return out
