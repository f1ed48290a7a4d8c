"""Minimal BCF2 (BGZF-compressed binary VCF) reader — enough for varlociraptor's observation and calls files.

Spec: VCFv4.2 / BCF2.2 (samtools/hts-specs): BGZF = concatenated gzip members; magic `BCF\\2\\2`; header text;
records = (l_shared, l_indiv, CHROM, POS, rlen, QUAL, n_allele<<16|n_info, n_fmt<<24|n_sample, ID, alleles,
FILTER, INFO key/value pairs, FORMAT blocks) with typed values (descriptor byte len<<4|type; types 1/2/3 ints,
5 float, 7 char).  Replaces rust-htslib's bcf::Reader for this path (reference calling.rs:306-318); no htslib here.
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
            yield {"chrom": self.contigs.get(chrom, str(chrom)), "pos": pos + 1, "id": bytes(rid).decode() if rid else ".",
                   "ref": alleles[0] if alleles else ".", "alt": ",".join(alleles[1:]) if len(alleles) > 1 else ".", "info": info}
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
        recs.append((rec["chrom"], rec["pos"], rec["ref"], rec["alt"], info))
    return r.header_lines, recs
