"""PYTHON CROSS-CHECK of the native v15 decoder (csrc/vlr_ingest.cpp on the host, csrc/vlr_decode.hip on the device), not the
product path: the tests compare the native readers with it record by record, and `VLR_INGEST=python` selects it in the CLI.

Decoder for varlociraptor's observation format v15 (text VCF flavour) -> PileupBatch.

Wire format (src/calling/variants/preprocessing/mod.rs:810-1038): every per-observation vector is
`bincode(Vec<T>)` (little-endian, u64 length prefix, u32 enum variant index, u8 Option tag), split
into LE u16 words, each stored as one i32 of an INFO integer vector (odd byte counts zero-padded,
mod.rs:985-988).  T = MiniLogProb{F16(f16) | F32(f32)} (src/utils/mod.rs:449-474), Option<…>,
C-like enums, bv::BitVec<u8> = {Option tag, u64 nblocks, blocks, u64 nbits}.

Text `.vcf` and binary `.bcf` (via the pure-Python reader in bcfio.py) are accepted; the matching cross-check of the calls
writer is callsfmt.py.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import abi
from .batch import PileupBatch

OBSERVATION_FORMAT_VERSION = "15"  # preprocessing/mod.rs:810


def _bytes_from_info(values: str) -> bytes:
    words = np.array([int(v) for v in values.split(",")], dtype=np.int64).astype(np.uint16)
    return words.astype("<u2").tobytes()


class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        self.i = 0

    def u8(self):
        v = self.b[self.i]
        self.i += 1
        return v

    def i8(self):
        return struct.unpack_from("<b", self.b, self._adv(1))[0]

    def _adv(self, n):
        i = self.i
        self.i += n
        return i

    def u32(self):
        return struct.unpack_from("<I", self.b, self._adv(4))[0]

    def u64(self):
        return struct.unpack_from("<Q", self.b, self._adv(8))[0]

    def minilogprob(self) -> float:
        tag = self.u32()
        if tag == 0:
            return float(np.frombuffer(self.b, dtype="<f2", count=1, offset=self._adv(2))[0])
        if tag == 1:
            return float(struct.unpack_from("<f", self.b, self._adv(4))[0])
        raise ValueError("invalid MiniLogProb variant %d" % tag)


def _vec_minilogprob(buf: bytes) -> np.ndarray:
    r = _Reader(buf)
    n = r.u64()
    # every element is a u32 variant tag + an f16 (6 bytes) or an f32 (8 bytes): a vector of n elements that is 6n or 8n bytes
    # long holds one variant only and decodes as a strided array; mixed vectors take the element loop
    body = len(buf) - 8
    if n and body == 6 * n:
        a = np.frombuffer(buf, dtype=np.dtype([("tag", "<u4"), ("v", "<f2")]), count=n, offset=8)
        if not a["tag"].any():
            return a["v"].astype(np.float32)
    elif n and body == 8 * n:
        a = np.frombuffer(buf, dtype=np.dtype([("tag", "<u4"), ("v", "<f4")]), count=n, offset=8)
        if (a["tag"] == 1).all():
            return a["v"].astype(np.float32)
    return np.array([r.minilogprob() for _ in range(n)], dtype=np.float32)


def _vec_opt_minilogprob(buf: bytes) -> np.ndarray:
    r = _Reader(buf)
    n = r.u64()
    out = np.full(n, np.nan, dtype=np.float32)
    for i in range(n):
        if r.u8():
            out[i] = r.minilogprob()
    return out


def _vec_enum(buf: bytes) -> np.ndarray:
    r = _Reader(buf)
    n = r.u64()
    return np.frombuffer(buf, dtype="<u4", count=n, offset=8).astype(np.uint32)


def _vec_opt_i8(buf: bytes) -> np.ndarray:
    r = _Reader(buf)
    n = r.u64()
    out = np.full(n, -128, dtype=np.int16)  # -128 = None
    for i in range(n):
        if r.u8():
            out[i] = r.i8()
    return out


def _vec_opt_u32(buf: bytes) -> np.ndarray:
    r = _Reader(buf)
    n = r.u64()
    out = np.full(n, -1, dtype=np.int64)  # -1 = None
    for i in range(n):
        if r.u8():
            out[i] = r.u32()
    return out


def _bitvec(buf: bytes) -> np.ndarray:
    r = _Reader(buf)
    if not r.u8():
        nbits = r.u64()
        return np.zeros(nbits, dtype=bool)
    nblocks = r.u64()
    blocks = np.frombuffer(buf, dtype=np.uint8, count=nblocks, offset=r._adv(nblocks))
    nbits = r.u64()
    bits = np.unpackbits(blocks, bitorder="little")[:nbits]
    return bits.astype(bool)


# bio_types::sequence::SequenceReadPairOrientation discriminants (F1R2 0, F2R1 1, …, None 8; the
# fixture pins None = 8) -> VLR_ORIENT_*
def _orient_class(v: np.ndarray) -> np.ndarray:
    out = np.full(v.shape, abi.ORIENT_OTHER, dtype=np.uint32)
    out[v == 0] = abi.ORIENT_F1R2
    out[v == 1] = abi.ORIENT_F2R1
    out[v == 8] = abi.ORIENT_NONE
    return out


def decode_record_info(info: Dict[str, str]) -> Dict[str, np.ndarray]:
    """One record's INFO map -> observation columns (read_observations, preprocessing/mod.rs:818-919)."""
    g = lambda k: _bytes_from_info(info[k])
    cols = {
        "prob_mapping": _vec_minilogprob(g("PROB_MAPPING")),
        "prob_ref": _vec_minilogprob(g("PROB_REF")),
        "prob_alt": _vec_minilogprob(g("PROB_ALT")),
        "prob_missed_allele": _vec_minilogprob(g("PROB_MISSED_ALLELE")),
        "prob_sample_alt": _vec_minilogprob(g("PROB_SAMPLE_ALT")),
        "prob_double_overlap": _vec_minilogprob(g("PROB_DOUBLE_OVERLAP")),
        "prob_hit_base": _vec_minilogprob(g("PROB_HIT_BASE")),
    }
    n = len(cols["prob_mapping"])
    strand = _vec_enum(g("STRAND"))
    orient = _orient_class(_vec_enum(g("READ_ORIENTATION")))
    readpos = _vec_enum(g("READ_POSITION"))  # 0 Major, 1 Some
    alt_locus = _vec_enum(g("ALT_LOCUS"))
    softclipped = _bitvec(g("SOFTCLIPPED"))
    paired = _bitvec(g("PAIRED"))
    max_mapq = _bitvec(g("IS_MAX_MAPQ"))
    is_hp = "PROB_HOMOPOLYMER_ARTIFACT_OBSERVABLE" in info  # mod.rs:867 is_homopolymer_indel
    if is_hp:
        cols["prob_hp_artifact"] = _vec_opt_minilogprob(g("PROB_HOMOPOLYMER_ARTIFACT_OBSERVABLE"))
        cols["prob_hp_variant"] = _vec_opt_minilogprob(g("PROB_HOMOPOLYMER_VARIANT_OBSERVABLE"))
        hp_len = _vec_opt_i8(g("HOMOPOLYMER_INDEL_LEN"))
    else:
        cols["prob_hp_artifact"] = np.full(n, np.nan, np.float32)
        cols["prob_hp_variant"] = np.full(n, np.nan, np.float32)
        hp_len = np.full(n, -128, np.int16)
    for a in (strand, orient, readpos, alt_locus, softclipped, paired, max_mapq):
        assert len(a) == n
    cols["flags"] = abi.pack_flags(strand, orient, readpos == 0, softclipped, paired, max_mapq, alt_locus, hp_len)
    cols["_is_homopolymer_indel"] = np.array([is_hp])
    # output-only feature (OBS string of the calls record, calling/variants/mod.rs:283-287)
    cols["_third_allele_evidence"] = _vec_opt_u32(g("THIRD_ALLELE_EVIDENCE")) if "THIRD_ALLELE_EVIDENCE" in info else np.full(n, -1, np.int64)
    return cols


def _variant_class(ref: str, alt: str) -> Tuple[int, bool, bool]:
    """(vlr_variant_type, is_snv_or_mnv, has_snv) following calling.rs:517-534 / collect_variants."""
    if alt.startswith("<"):
        t = alt.strip("<>")
        vt = {"DEL": abi.VT_INDEL, "INS": abi.VT_INDEL, "INV": abi.VT_SV, "DUP": abi.VT_SV, "BND": abi.VT_SV}.get(t, abi.VT_OTHER)
        # calling.rs compares allele byte lengths: "CG" vs "<METH>" differ => not snv/mnv
        return vt, len(ref) == len(alt), False
    if "[" in alt or "]" in alt:
        return abi.VT_SV, len(ref) == len(alt), False
    if len(ref) == 1 and len(alt) == 1:
        return abi.VT_SNV, True, True
    if len(ref) == len(alt):
        return abi.VT_MNV, True, False
    return abi.VT_INDEL, False, False


def _phred_info(info: Dict[str, str], key: str) -> Optional[float]:
    """Variant-specific prior from the candidate record (calling.rs:470-494): PHRED float -> LogProb; None if absent/missing."""
    v = info.get(key)
    if v is None or v in ("", "."):
        return None
    x = float(np.float32(float(v.split(",")[0])))  # INFO floats are f32 in the record (htslib), also when parsed from text
    if x != x:
        return None
    return -x * np.log(10.0) / 10.0


def haplotype_identifier(info: Dict[str, str]) -> Optional[str]:
    """HaplotypeIdentifier::from (variants/model/mod.rs:87-133): INFO EVENT, else the sorted pair (record ID, MATEID)."""
    ev = info.get("EVENT")
    if ev:
        return ev.split(",")[0]
    mate = info.get("MATEID")
    if mate:
        rid = info.get("__ID", ".")
        if rid in (".", ""):
            raise ValueError("breakend with MATEID but without record ID")  # errors::Error::BreakendMateidWithoutRecid
        return "-".join(sorted([rid, mate.split(",")[0]]))
    return None


def haplotype_groups(haplotypes: List[Optional[str]]) -> Tuple[List[int], List[int]]:
    """Breakends of one event share a pileup and a result (calling.rs:569-580, 726-741, 820-839): returns
    (representatives, source) where `representatives` are the loci to evaluate and source[l] indexes into them."""
    reps: List[int] = []
    first: Dict[str, int] = {}
    source: List[int] = []
    for l, h in enumerate(haplotypes):
        if h is not None and h in first:
            source.append(first[h])
            continue
        if h is not None:
            first[h] = len(reps)
        source.append(len(reps))
        reps.append(l)
    return reps, source


def read_observation_vcf(paths: List[str], omit_bias_mask: int = 0) -> Tuple[PileupBatch, List[Tuple[str, int, str, str]]]:
    """Read one observation VCF per sample (in sample-index order) into a PileupBatch.

    `omit_bias_mask`: VLR_BIAS_* bits of the `--omit-*` flags (calling.rs:63-68).  locus_flags follow
    WorkItem.check_* (calling.rs:557-567) with every record treated as precise unless INFO has IMPRECISE.
    """
    per_sample = []
    for path in paths:
        recs = []
        version_ok = False
        if path.endswith(".bcf"):
            from .bcfio import bcf_to_vcf_info_records
            hdr, recs = bcf_to_vcf_info_records(path)
            version_ok = any(l.strip() == "##varlociraptor_observation_format_version=" + OBSERVATION_FORMAT_VERSION for l in hdr)
            if not version_ok:
                raise ValueError("invalid observation format (calling.rs:324-339)")
            per_sample.append(recs)
            continue
        with open(path) as fh:
            for line in fh:
                if line.startswith("##varlociraptor_observation_format_version="):
                    version_ok = line.strip().split("=", 1)[1] == OBSERVATION_FORMAT_VERSION
                if line.startswith("#"):
                    continue
                f = line.rstrip("\n").split("\t")
                info = {}
                for kv in f[7].split(";"):
                    if "=" in kv:
                        k, v = kv.split("=", 1)
                        info[k] = v
                    else:
                        info[kv] = ""
                info["__ID"] = f[2]
                recs.append((f[0], int(f[1]), f[3], f[4], info))
        if not version_ok:
            raise ValueError("invalid observation format (calling.rs:324-339)")  # errors::Error::InvalidObservationFormat
        per_sample.append(recs)
    n = len(per_sample[0])
    for recs in per_sample[1:]:
        if len(recs) != n:
            raise ValueError("inconsistent observations (calling.rs:369-371)")
    S = len(paths)
    offsets = [0]
    cols: Dict[str, List[np.ndarray]] = {k: [] for k, _ in abi.OBS_COLUMNS}
    third: List[np.ndarray] = []
    locus_flags, vtypes, refb, altb, sites, haplotypes, priors = [], [], [], [], [], [], []
    for i in range(n):
        chrom, pos, ref, alt, _ = per_sample[0][i]
        for recs in per_sample[1:]:
            if recs[i][:4] != (chrom, pos, ref, alt):
                raise ValueError("inconsistent observations (calling.rs:379-390)")
        vt, snv_or_mnv, has_snv = _variant_class(ref, alt)
        precise = "IMPRECISE" not in per_sample[0][i][4]
        any_hp = False
        for s in range(S):
            c = decode_record_info(per_sample[s][i][4])
            any_hp |= bool(c.pop("_is_homopolymer_indel")[0])
            third.append(c.pop("_third_allele_evidence"))
            for k, _ in abi.OBS_COLUMNS:
                cols[k].append(c[k])
            offsets.append(offsets[-1] + len(c["prob_mapping"]))
        m = 0
        if snv_or_mnv and precise and not (omit_bias_mask & abi.BIAS_ORIENTATION):
            m |= abi.BIAS_ORIENTATION
        if precise and not (omit_bias_mask & abi.BIAS_STRAND):
            m |= abi.BIAS_STRAND
        if snv_or_mnv and precise and not (omit_bias_mask & abi.BIAS_POSITION):
            m |= abi.BIAS_POSITION
        if snv_or_mnv and precise and not (omit_bias_mask & abi.BIAS_SOFTCLIP):
            m |= abi.BIAS_SOFTCLIP
        if any_hp and not (omit_bias_mask & abi.BIAS_HOMOPOLYMER):
            m |= abi.BIAS_HOMOPOLYMER
        if not (omit_bias_mask & abi.BIAS_ALTLOCUS):
            m |= abi.BIAS_ALTLOCUS
        if snv_or_mnv and not (omit_bias_mask & abi.BIAS_ORIENTATION):
            m |= abi.LOCUS_REMOVE_NONSTANDARD  # calling.rs:590-598
        if has_snv:
            m |= abi.LOCUS_HAS_SNV
        locus_flags.append(m)
        vtypes.append(vt)
        refb.append(ord(ref[0]) if has_snv else 0)
        altb.append(ord(alt[0]) if has_snv else 0)
        sites.append((chrom, pos, ref, alt))
        haplotypes.append(haplotype_identifier(per_sample[0][i][4]))
        priors.append(tuple(_phred_info(per_sample[0][i][4], k) for k in ("HETEROZYGOSITY", "SOMATIC_EFFECTIVE_MUTATION_RATE")))
    columns = {k: (np.concatenate(v) if v else np.zeros(0, dt)) for (k, dt), v in zip(abi.OBS_COLUMNS, cols.values())}
    locus = {"locus_flags": np.array(locus_flags, np.uint8), "variant_type": np.array(vtypes, np.uint8),
             "ref_base": np.array(refb, np.uint8), "alt_base": np.array(altb, np.uint8)}
    batch = PileupBatch(S, np.array(offsets, np.uint32), columns, locus)
    batch.extra = {"third_allele_evidence": np.concatenate(third) if third else np.zeros(0, np.int64), "haplotype": haplotypes,
                   "prior_overrides": priors}
    return batch, sites
