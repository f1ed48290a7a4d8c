"""PYTHON CROSS-CHECK of the native calls writer (`vlr_calls_write` / `vlr_calls_writer_*`, csrc/vlr_ingest.cpp), not the product
path: the end-to-end tests require the native writer's files to equal this formatter's byte for byte.

Calls record formatting: the text-VCF flavour of Call::write_final_record
(reference src/calling/variants/mod.rs:178-600) for records evaluated by the engine.

Produces, per record, INFO `PROB_<EVENT>` (PHRED, f32, sorted by descending probability, mod.rs:223-231,
447-466) and FORMAT `DP:AF:SAOBS:SROBS:OBS:OOBS:SB:ROB:RPB:SCB:HE:ALB:AFD` (mod.rs:233-559).
BCF output of the same records goes through bcfio.BcfWriter (cross-check) or the native writer (product); SURVEY.md §8(f)#2.
"""
from __future__ import annotations

from collections import Counter
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import abi
from .batch import CallResults, PileupBatch

FORMAT_KEYS = "DP:AF:SAOBS:SROBS:OBS:OOBS:SB:ROB:RPB:SCB:HE:ALB:AFD"
LN10 = np.log(10.0)


def fmt_float(x: float) -> str:
    """htslib's VCF float formatting (kputd: 6 significant digits, trailing zeros stripped) of an f32."""
    x = float(np.float32(x))
    if np.isnan(x):
        return "."
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    return "%g" % x


def _relative_eq(a: float, b: float) -> bool:
    if a == b:
        return True
    d = abs(a - b)
    eps = np.finfo(np.float64).eps
    return d <= eps or d <= max(abs(a), abs(b)) * eps


def bayes_factor_to_letter(bf: float) -> str:
    """utils/mod.rs:158-167 over bio's Kass-Raftery scale (<=1 None, <=3 Barely, <=20 Positive, <=150 Strong)."""
    if bf <= 1.0:
        return "E" if _relative_eq(bf, 1.0) else "N"
    if bf <= 3.0:
        return "B"
    if bf <= 20.0:
        return "P"
    if bf <= 150.0:
        return "S"
    return "V"


def generalized_cigar(items: Sequence[str], aux_sort) -> str:
    """utils/mod.rs:122-156 with keep_order = false: Counter::most_common then a stable sort by `aux_sort`."""
    c = Counter(items)
    mc = sorted(c.items(), key=lambda kv: -kv[1])  # ties: unspecified in the reference
    mc = sorted(mc, key=aux_sort)
    return "".join("%d%s" % (n, it) for it, n in mc)


def _kept_rows(batch: PileupBatch, l: int, s: int) -> np.ndarray:
    sl = batch.pileup_slice(l, s)
    idx = np.arange(sl.start, sl.stop)
    if batch.locus["locus_flags"][l] & abi.LOCUS_REMOVE_NONSTANDARD:  # pileup.rs:26-43
        orient = (batch.columns["flags"][idx] >> abi.F_ORIENT_SHIFT) & 3
        idx = idx[orient != abi.ORIENT_OTHER]
    return idx


def sample_fields(batch: PileupBatch, res: CallResults, l: int, s: int) -> List[str]:
    """FORMAT values of one sample (mod.rs:233-360, 473-559)."""
    sl = batch.pileup_slice(l, s)
    idx = _kept_rows(batch, l, s)
    n_filtered = (sl.stop - sl.start) - len(idx)
    pa = batch.columns["prob_alt"][idx].astype(np.float64)
    pr = batch.columns["prob_ref"][idx].astype(np.float64)
    pm = batch.columns["prob_mapping"][idx].astype(np.float64)
    fl = batch.columns["flags"][idx]
    third = getattr(batch, "extra", {}).get("third_allele_evidence")
    third = third[idx] if third is not None else np.full(len(idx), -1)
    # expected_depth (read_observation.rs:43-47)
    dp = int(np.floor(np.exp(pm).sum() + 0.5)) if len(idx) else 0
    obs_items, alt_items, ref_items = [], [], []
    with np.errstate(over="ignore", invalid="ignore"):
        bf_alt = np.exp(pa - pr)
        bf_ref = np.exp(pr - pa)
    for i in range(len(idx)):
        f = int(fl[i])
        maxq = bool(f & abi.F_MAX_MAPQ)
        if bf_alt[i] > bf_ref[i]:
            score = "A" + bayes_factor_to_letter(bf_alt[i])
        elif bf_ref[i] > bf_alt[i]:
            score = "R" + bayes_factor_to_letter(bf_ref[i])
        else:
            score = "E"
        score = score.upper() if maxq else score.lower()
        strand = (f >> abi.F_STRAND_SHIFT) & 3
        orient = (f >> abi.F_ORIENT_SHIFT) & 3
        altloc = (f >> abi.F_ALTLOCUS_SHIFT) & 3
        hp_err = bool(f & abi.F_HP_LEN_VALID) and ((f >> abi.F_HP_LEN_SHIFT) & 0xFF) != 0
        obs_items.append("%s%s%s%s%s%s%s%s%s" % (
            score, ("%d" % third[i]) if third[i] >= 0 else ".", "p" if f & abi.F_PAIRED else "s",
            "#*."[altloc], "+-*."[strand], "><*!"[orient], "^" if f & abi.F_READPOS_MAJOR else "*",
            "$" if f & abi.F_SOFTCLIPPED else ".", "*" if hp_err else "."))
        if pa[i] > pr[i]:
            letter = bayes_factor_to_letter(bf_alt[i])
            alt_items.append(letter.upper() if maxq else letter.lower())
        else:
            letter = bayes_factor_to_letter(bf_ref[i])
            ref_items.append(letter.upper() if maxq else letter.lower())
    obs = generalized_cigar(obs_items, lambda kv: 2 if kv[0].startswith("N") else (1 if kv[0].startswith("E") else 0))
    simple_key = lambda kv: 2 if kv[0].startswith("R") else (1 if kv[0].endswith("E") else 0)
    saobs = generalized_cigar(alt_items, simple_key)
    srobs = generalized_cigar(ref_items, simple_key)
    mb = res.map_bias[l]
    sym = [".+-"[mb[0]], ".><"[mb[1]], ".^"[mb[2]], ".$"[mb[3]], ".*"[mb[4]], ".*"[mb[5]]]
    afd = "."
    if res.afd_count is not None and not mb.any():
        n = min(int(res.afd_count[l, s]), res.afd_capacity)
        v = res.afd_vaf[l, s, :n]
        p = res.afd_lnprob[l, s, :n]
        order = np.argsort(v, kind="stable")
        with np.errstate(invalid="ignore"):
            ph = -10.0 * p / LN10
        afd = ",".join("%.3f=%.2f" % (v[i], ph[i] + 0.0) for i in order) if n else ""
    return [str(dp), fmt_float(res.map_vaf[l, s]), saobs or ".", srobs or ".", obs or ".", str(n_filtered)] + sym + [afd]


def format_record(site: Tuple[str, int, str, str], batch: PileupBatch, res: CallResults, l: int, out_names: Sequence[str],
                  sample_names: Sequence[str]) -> str:
    """One VCF line of the calls file for locus l."""
    chrom, pos, ref, alt = site
    missing = bool(res.status[l] & abi.LOCUS_MISSING_DATA)
    probs = [(name, res.ln_posterior[l, i]) for i, name in enumerate(out_names)]
    probs.sort(key=lambda kv: -kv[1] if kv[1] == kv[1] else np.inf)  # mod.rs:231 sort by descending probability
    info = []
    for name, lp in probs:
        tag = "PROB_" + name.upper()
        if missing:
            info.append("%s=." % tag)
        else:
            info.append("%s=%s" % (tag, fmt_float(abs(-10.0 * lp / LN10))))
    if missing:
        S = len(sample_names)
        samples = [":".join(["0", "."] + ["."] * 11)] * S
    else:
        samples = [":".join(sample_fields(batch, res, l, s)) for s in range(len(sample_names))]
    return "\t".join([chrom, str(pos), ".", ref, alt, ".", ".", ";".join(info), FORMAT_KEYS] + samples)


def header(out_names: Sequence[str], sample_names: Sequence[str], contigs: Sequence[str] = ()) -> str:
    """Header lines the reference adds (calling.rs:94-294), abridged descriptions."""
    h = ["##fileformat=VCFv4.2"]
    for c in contigs:
        h.append("##contig=<ID=%s>" % c)
    for name in out_names:
        h.append('##INFO=<ID=PROB_%s,Number=A,Type=Float,Description="Posterior probability for event %s (PHRED)">' % (name.upper(), name))
    h.append('##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Expected sequencing depth, while considering mapping uncertainty">')
    h.append('##FORMAT=<ID=AF,Number=A,Type=Float,Description="Maximum a posteriori probability estimate of allele frequency">')
    for k in ("SAOBS", "SROBS", "OBS", "SB", "ROB", "RPB", "SCB", "HE", "ALB"):
        h.append('##FORMAT=<ID=%s,Number=A,Type=String,Description="see varlociraptor">' % k)
    h.append('##FORMAT=<ID=OOBS,Number=A,Type=Integer,Description="Number of omitted observations">')
    h.append('##FORMAT=<ID=AFD,Number=.,Type=String,Description="Sampled posterior probability densities of allele frequencies in PHRED scale">')
    h.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(sample_names))
    return "\n".join(h)
