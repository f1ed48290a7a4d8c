"""PYTHON CROSS-CHECK of the native BGZF / BCF2 code (csrc/vlr_ingest.cpp, csrc/vlr_inflate.hip), not the product path: used by
the tests and by `VLR_INGEST=python`.

Minimal BCF2 (BGZF-compressed binary VCF) reader and writer — enough for varlociraptor's observation and calls files.

Spec: VCFv4.2 / BCF2.2 (samtools/hts-specs): BGZF = concatenated gzip members; magic `BCF\\2\\2`; header text;
records = (l_shared, l_indiv, CHROM, POS, rlen, QUAL, n_allele<<16|n_info, n_fmt<<24|n_sample, ID, alleles,
FILTER, INFO key/value pairs, FORMAT blocks) with typed values (descriptor byte len<<4|type; types 1/2/3 ints,
5 float, 7 char).  Restates what rust-htslib's bcf::Reader / Writer do for this path (reference calling.rs:296-318); the product path is
`vlr_obs_read` / `vlr_obs_reader_open(_device)` / `vlr_calls_write` behind the C ABI.
"""
from __future__ import annotations

import gzip
import re
import struct
from typing import Dict, Iterator, List, Tuple

INT_MISSING = {1: -128, 2: -32768, 3: -2147483648}
INT_EOV = {1: -127, 2: -32767, 3: -2147483647}
_FMT = {1: "b", 2: "h", 3: "i", 5: "f"}
_SIZE = {1: 1, 2: 2, 3: 4, 5: 4, 7: 1}


class BcfReader:
    def __init__(self, path: str):
        with gzip.open(path, "rb") as fh:
            self.buf = fh.read()
        if self.buf[:5] != b"BCF\x02\x02":
            raise ValueError("not a BCF2.2 file: %r" % path)
        l_text = struct.unpack_from("<I", self.buf, 5)[0]
        self.header_text = self.buf[9:9 + l_text].rstrip(b"\x00").decode()
        self.pos = 9 + l_text
        self.header_lines = [l for l in self.header_text.split("\n") if l.startswith("##")]
        self.samples = []
        for l in self.header_text.split("\n"):
            if l.startswith("#CHROM"):
                f = l.split("\t")
                self.samples = f[9:]
        # dictionaries (BCF2.2 §6.2.1): strings = FILTER/INFO/FORMAT IDs in order of appearance, PASS = 0; contigs
        self.strings: Dict[int, str] = {0: "PASS"}
        self.contigs: Dict[int, str] = {}
        seen = {"PASS": 0}
        nxt, cnxt = 1, 0
        for l in self.header_lines:
            m = re.match(r"##(FILTER|INFO|FORMAT)=<(.*)>", l)
            if m:
                idm = re.search(r"(?:^|,)ID=([^,>]+)", m.group(2))
                idx = re.search(r"(?:^|,)IDX=(\d+)", m.group(2))
                name = idm.group(1)
                if name in seen:
                    continue
                i = int(idx.group(1)) if idx else nxt
                seen[name] = i
                self.strings[i] = name
                nxt = max(nxt, i + 1)
            m = re.match(r"##contig=<(.*)>", l)
            if m:
                idm = re.search(r"(?:^|,)ID=([^,>]+)", m.group(1))
                idx = re.search(r"(?:^|,)IDX=(\d+)", m.group(1))
                i = int(idx.group(1)) if idx else cnxt
                self.contigs[i] = idm.group(1)
                cnxt = max(cnxt, i + 1)

    # ---- typed values
    def _typed_desc(self, p: int) -> Tuple[int, int, int]:
        b = self.buf[p]
        p += 1
        n, t = b >> 4, b & 0xF
        if n == 15:
            vals, p = self._typed(p)
            n = vals[0]
        return n, t, p

    def _typed(self, p: int):
        n, t, p = self._typed_desc(p)
        if t == 0 or n == 0:
            return [], p
        if t == 7:
            s = self.buf[p:p + n]
            return s, p + n
        vals = list(struct.unpack_from("<%d%s" % (n, _FMT[t]), self.buf, p))
        if t in INT_EOV:
            vals = [v for v in vals if v != INT_EOV[t]]
            vals = [None if v == INT_MISSING[t] else v for v in vals]
        return vals, p + n * _SIZE[t]

    def __iter__(self) -> Iterator[dict]:
        p = self.pos
        buf = self.buf
        while p + 8 <= len(buf):
            l_shared, l_indiv = struct.unpack_from("<II", buf, p)
            rec_start = p
            p += 8
            end = p + l_shared + l_indiv
            chrom, pos, rlen, qual, nai, nfs = struct.unpack_from("<iiifII", buf, p)
            q = p + 24
            n_allele, n_info = nai >> 16, nai & 0xFFFF
            rid, q = self._typed(q)
            alleles = []
            for _ in range(n_allele):
                a, q = self._typed(q)
                alleles.append(bytes(a).decode())
            filt, q = self._typed(q)
            info: Dict[str, list] = {}
            for _ in range(n_info):
                key, q = self._typed(q)
                val, q = self._typed(q)
                if isinstance(val, (bytes, bytearray)):
                    val = bytes(val).decode()
                info[self.strings[key[0]]] = val
            n_fmt, n_sample = nfs >> 24, nfs & 0xFFFFFF
            fmt: Dict[str, list] = {}
            for _ in range(n_fmt):
                key, q = self._typed(q)
                n, t, q = self._typed_desc(q)
                per_sample = []
                for _s in range(n_sample):
                    if t == 7:
                        per_sample.append(bytes(buf[q:q + n]).rstrip(b"\x00").decode())
                        q += n
                    else:
                        vals = list(struct.unpack_from("<%d%s" % (n, _FMT[t]), buf, q)) if n else []
                        q += n * _SIZE.get(t, 0)
                        if t in INT_EOV:
                            vals = [v for v in vals if v != INT_EOV[t]]
                            vals = [None if v == INT_MISSING[t] else v for v in vals]
                        per_sample.append(vals)
                fmt[self.strings[key[0]]] = per_sample
            yield {"chrom": self.contigs.get(chrom, str(chrom)), "pos": pos + 1, "id": bytes(rid).decode() if rid else ".",
                   "ref": alleles[0] if alleles else ".", "alt": ",".join(alleles[1:]) if len(alleles) > 1 else ".", "info": info,
                   "qual": qual, "filter": [self.strings[i] for i in filt] if filt else [], "format": fmt,
                   "raw": bytes(buf[rec_start:end])}  # the encoded record, for writers that pass records through unchanged
            p = end


def bcf_to_vcf_info_records(path: str) -> Tuple[List[str], List[Tuple[str, int, str, str, Dict[str, str]]]]:
    """Records of a BCF in the shape obsfmt uses for text VCFs: INFO values as comma-joined strings."""
    r = BcfReader(path)
    recs = []
    for rec in r:
        info = {}
        for k, v in rec["info"].items():
            if isinstance(v, str):
                info[k] = v
            elif isinstance(v, list):
                info[k] = ",".join("." if x is None else ("%d" % x if isinstance(x, int) else repr(x)) for x in v)
            else:
                info[k] = ""
        info["__ID"] = rec["id"]
        recs.append((rec["chrom"], rec["pos"], rec["ref"], rec["alt"], info))
    return r.header_lines, recs


# ------------------------------------------------------------------------------------------------------------------
# writer: text VCF (header lines + record lines) -> BCF2.2 in BGZF blocks.  Stands in for rust-htslib's bcf::Writer
# on the calls side of the process boundary (reference calling/variants/calling.rs:296-304, mod.rs:447-600).
import zlib

FLOAT_MISSING = 0x7F800001
FLOAT_EOV = 0x7F800002
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25  # header 18 + trailer 8, minus 1
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _desc(n: int, t: int) -> bytes:
    if n < 15:
        return bytes([(n << 4) | t])
    return bytes([0xF0 | t]) + _enc_ints([n])


def _int_type(vals) -> int:
    lo = min((v for v in vals if v is not None), default=0)
    hi = max((v for v in vals if v is not None), default=0)
    if -120 <= lo and hi <= 127:
        return 1
    if -32760 <= lo and hi <= 32767:
        return 2
    return 3


def _enc_ints(vals, pad_to: int = 0, t: int = 0) -> bytes:
    t = t or _int_type(vals)
    out = [INT_MISSING[t] if v is None else v for v in vals]
    out += [INT_EOV[t]] * max(0, pad_to - len(out))
    return _desc(len(out), t) + struct.pack("<%d%s" % (len(out), _FMT[t]), *out)


def _f32_bits(v) -> int:
    if v is None:
        return FLOAT_MISSING
    return struct.unpack("<I", struct.pack("<f", v))[0]


def _enc_floats(vals, pad_to: int = 0) -> bytes:
    bits = [_f32_bits(v) for v in vals] + [FLOAT_EOV] * max(0, pad_to - len(vals))
    return _desc(len(bits), 5) + struct.pack("<%dI" % len(bits), *bits)


def _enc_str(sv: str) -> bytes:
    b = sv.encode()
    return _desc(len(b), 7) + b


def _parse_float(tok: str):
    if tok == ".":
        return None
    return float(tok)


class BcfWriter:
    """Encodes text-VCF records against their header.  Value types come from the header's Type= attributes;
    integers take the smallest width that holds a vector (BCF2.2 §6.3.3), floats are f32, Flags an INT8 1."""

    def __init__(self, path: str, header_text: str):
        lines = [l for l in header_text.split("\n") if l]
        self.meta = [l for l in lines if l.startswith("##")]
        self.chrom_line = [l for l in lines if l.startswith("#CHROM")][0]
        self.samples = self.chrom_line.split("\t")[9:]
        self.types: Dict[Tuple[str, str], str] = {}
        self.strings: Dict[str, int] = {"PASS": 0}
        self.contigs: Dict[str, int] = {}
        has_pass = any(l.startswith("##FILTER=<ID=PASS") for l in self.meta)
        if not has_pass:
            self.meta.insert(1, '##FILTER=<ID=PASS,Description="All filters passed">')
        for l in self.meta:
            m = re.match(r"##(FILTER|INFO|FORMAT)=<(.*)>", l)
            if m:
                name = re.search(r"(?:^|,)ID=([^,>]+)", m.group(2)).group(1)
                ty = re.search(r"(?:^|,)Type=([^,>]+)", m.group(2))
                self.types[(m.group(1), name)] = ty.group(1) if ty else "Flag"
                idx = re.search(r"(?:^|,)IDX=(\d+)", m.group(2))
                if name not in self.strings:
                    self.strings[name] = int(idx.group(1)) if idx else (max(self.strings.values()) + 1)
            m = re.match(r"##contig=<(.*)>", l)
            if m:
                name = re.search(r"(?:^|,)ID=([^,>]+)", m.group(1)).group(1)
                idx = re.search(r"(?:^|,)IDX=(\d+)", m.group(1))
                self.contigs.setdefault(name, int(idx.group(1)) if idx else len(self.contigs))
        text = ("\n".join(self.meta + [self.chrom_line]) + "\n").encode() + b"\x00"
        self.fh = open(path, "wb")
        self.pending = bytearray(b"BCF\x02\x02" + struct.pack("<I", len(text)) + text)

    def _flush(self, force: bool = False):
        while len(self.pending) >= 0xFF00 or (force and self.pending):
            chunk = bytes(self.pending[:0xFF00])
            del self.pending[:0xFF00]
            self.fh.write(_bgzf_block(chunk))

    def _enc_value(self, ty: str, text: str, pad_to: int = 0) -> bytes:
        if ty == "Integer":
            return _enc_ints([None if t == "." else int(t) for t in text.split(",")], pad_to)
        if ty == "Float":
            return _enc_floats([_parse_float(t) for t in text.split(",")], pad_to)
        if ty == "Flag":
            return b"\x11\x01"
        return _enc_str(text)

    def write_line(self, line: str):
        f = line.rstrip("\n").split("\t")
        chrom, pos, rid, ref, alt, qual, filt, info = f[:8]
        if chrom not in self.contigs:
            raise ValueError("contig %r not in the header" % chrom)
        alleles = [ref] + ([a for a in alt.split(",")] if alt != "." else [])
        infos = [kv for kv in info.split(";") if kv and kv != "."]
        shared = bytearray()
        shared += _enc_str(rid if rid != "." else "")
        for a in alleles:
            shared += _enc_str(a)
        if filt in (".", ""):
            shared += b"\x00"
        else:
            shared += _enc_ints([self.strings[x] for x in filt.split(";")])
        for kv in infos:
            k, _, v = kv.partition("=")
            ty = self.types.get(("INFO", k))
            if ty is None:
                raise ValueError("INFO key %r not in the header" % k)
            shared += _enc_ints([self.strings[k]]) + self._enc_value(ty, v)
        indiv = bytearray()
        n_fmt = 0
        if len(f) > 9 and self.samples:
            keys = f[8].split(":")
            cols = [c.split(":") for c in f[9:9 + len(self.samples)]]
            n_fmt = len(keys)
            for ki, k in enumerate(keys):
                ty = self.types.get(("FORMAT", k))
                if ty is None:
                    raise ValueError("FORMAT key %r not in the header" % k)
                vals = [c[ki] if ki < len(c) else "." for c in cols]
                indiv += _enc_ints([self.strings[k]])
                if ty == "Integer":
                    vecs = [[None if t == "." else int(t) for t in v.split(",")] for v in vals]
                    n = max(len(v) for v in vecs)
                    t = _int_type([x for v in vecs for x in v])
                    indiv += _desc(n, t)
                    for v in vecs:
                        out = [INT_MISSING[t] if x is None else x for x in v] + [INT_EOV[t]] * (n - len(v))
                        indiv += struct.pack("<%d%s" % (n, _FMT[t]), *out)
                elif ty == "Float":
                    vecs = [[_parse_float(t) for t in v.split(",")] for v in vals]
                    n = max(len(v) for v in vecs)
                    indiv += _desc(n, 5)
                    for v in vecs:
                        bits = [_f32_bits(x) for x in v] + [FLOAT_EOV] * (n - len(v))
                        indiv += struct.pack("<%dI" % n, *bits)
                else:
                    bs = [v.encode() for v in vals]
                    n = max(len(b) for b in bs)
                    indiv += _desc(n, 7)
                    for b in bs:
                        indiv += b + b"\x00" * (n - len(b))
        rlen = len(ref)
        for kv in infos:  # END defines rlen for symbolic alleles (BCF2.2 §6.3.1)
            if kv.startswith("END="):
                rlen = int(kv[4:]) - int(pos) + 1
        qbits = FLOAT_MISSING if qual in (".", "") else _f32_bits(float(qual))
        fixed = struct.pack("<iiiIII", self.contigs[chrom], int(pos) - 1, rlen, qbits, (len(alleles) << 16) | len(infos),
                            (n_fmt << 24) | len(self.samples))
        self.pending += struct.pack("<II", len(fixed) + len(shared), len(indiv)) + fixed + shared + indiv
        self._flush()

    def write_raw(self, raw: bytes):
        """A record in its BCF encoding (BcfReader's rec["raw"]), valid against the same header (dictionary indices)."""
        self.pending += raw
        self._flush()

    def close(self):
        self._flush(force=True)
        self.fh.write(_BGZF_EOF)
        self.fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def vcf_to_bcf(vcf_path: str, bcf_path: str) -> int:
    """Re-encode a text VCF (plain or gzip) as BCF2 in BGZF members with BcfWriter; returns the record count."""
    import gzip
    op = gzip.open if open(vcf_path, "rb").read(2) == b"\x1f\x8b" else open
    with op(vcf_path, "rt") as fh:
        lines = fh.read().split("\n")
    head = [l for l in lines if l.startswith("#")]
    recs = [l for l in lines if l and not l.startswith("#")]
    with BcfWriter(bcf_path, "\n".join(head) + "\n") as w:
        for l in recs:
            w.write_line(l)
    return len(recs)
