"""Bayesian FDR control over varlociraptor calls ("next" row §8(f)#4).

Mirrors `filter-calls control-fdr` (reference src/filtration/fdr.rs:36-158, src/utils/mod.rs:169-374; Müller,
Parmigiani & Rice 2006): collect the posterior of the chosen events per variant, sort, expected FDR = running mean of
the posterior error probabilities, threshold at alpha, then filter.  The O(n log n) part — sort, PEP prefix sums and the
boundary search — has a HIP implementation behind the C ABI (`vlr_fdr_threshold`, csrc/vlr_fdr.hip; `device="cuda"`) next
to the host restatement used by the CPU suite; record typing follows utils/collect_variants.rs:44-304.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

LN10 = math.log(10.0)
NUMERICAL_EPSILON = 1e-3  # utils/mod.rs:40
LN_05 = math.log(0.5)


def variant_types(rec: dict) -> List[Optional[Tuple[str, int]]]:
    """(type, length) per ALT allele as collect_variants would type it; None = skipped."""
    info = rec["info"]
    ref = rec["ref"]
    alts = rec["alt"].split(",") if rec["alt"] != "." else []
    pos0 = rec["pos"] - 1
    svlens = [abs(x) if x is not None else None for x in info["SVLEN"]] if isinstance(info.get("SVLEN"), list) else None
    end = info["END"][0] - 1 if isinstance(info.get("END"), list) and info["END"] else None
    svtype = info.get("SVTYPE")
    valid_del = lambda r, a: a == "<DEL>" or (len(r) > len(a) and r[:len(a)] == a and len(a) == 1)
    valid_ins = lambda r, a: a == "<INS>" or (len(r) < len(a) and r == a[:len(r)] and len(r) == 1)
    out: List[Optional[Tuple[str, int]]] = []
    if svtype:
        if svtype in ("INV", "DUP"):
            out.append((svtype, end + 1 - pos0) if (len(alts) == 1 and end is not None) else None)
        elif svtype == "BND":
            out.extend(("BND", 0) for _ in alts)
        elif svtype == "INS":
            a = alts[0]
            out.append(("INS", len(a) - len(ref)) if (a != "<INS>" and valid_ins(ref, a)) else None)
        elif svtype == "DEL":
            if svlens and svlens[0] is not None:
                svlen = svlens[0]
            elif svlens is None and end is not None:
                svlen = end - (pos0 + 1)  # collect_variants.rs:196-199
            else:
                raise ValueError("missing SVLEN or END")
            a = alts[0]
            out.append(("DEL", svlen) if valid_del(ref, a) else None)
        return out
    for i, a in enumerate(alts):
        if a == "<*>":
            out.append(("REF", 0))
        elif a == "<DEL>":
            out.append(("DEL", svlens[i]) if (svlens and svlens[i] is not None) else None)
        elif a == "<METH>":
            out.append(("METH", 0))
        elif a.startswith("<"):
            out.append(None)
        elif len(a) == 1 and len(ref) == 1:
            out.append(("SNV", 1))
        elif len(a) == len(ref):
            out.append(("MNV", len(a)))
        elif valid_del(ref, a):
            out.append(("DEL", len(ref) - len(a)))
        elif valid_ins(ref, a):
            out.append(("INS", len(a) - len(ref)))
        else:
            out.append(("REP", 0))
    return out


def _is_type(v: Optional[Tuple[str, int]], vartype) -> bool:
    """Variant::is_type (variants/model/mod.rs); vartype = (kind, (lo, hi) | None) or None for all."""
    if v is None:
        return False
    if vartype is None:
        return True
    kind, rng = vartype
    if v[0] != kind:
        return False
    return rng is None or (rng[0] <= v[1] < rng[1])


def _lse(v: Sequence[float]) -> float:
    m = max(v)
    if m == -math.inf:
        return -math.inf
    return m + math.log(sum(math.exp(x - m) for x in v))


def tags_prob_sum(rec: dict, tags: Sequence[str], vartype) -> List[Optional[float]]:
    """utils/mod.rs:177-212: ln-sum over the given PROB_* tags per variant (PHRED in the file)."""
    variants = [v for v in variant_types(rec) if v is not None]
    acc: List[List[float]] = [[] for _ in variants]
    for tag in tags:
        vals = rec["info"].get(tag)
        if not isinstance(vals, list):
            continue
        for i, (v, p) in enumerate(zip(variants, vals)):
            if p is None or (isinstance(p, float) and math.isnan(p)) or not _is_type(v, vartype):
                continue
            acc[i].append(-float(p) * LN10 / 10.0)
    out = []
    for probs in acc:
        if probs:
            s = _lse(probs)
            if 0.0 < s <= NUMERICAL_EPSILON:  # cap_numerical_overshoot
                s = 0.0
            out.append(s)
        else:
            out.append(None)
    return out


def collect_prob_dist(records: Iterable[dict], tags: Sequence[str], vartype) -> List[float]:
    """utils/mod.rs:236-270 (ascending)."""
    seen = set()
    dist = []
    for rec in records:
        ev = rec["info"].get("EVENT")
        if isinstance(ev, str):
            if ev in seen:
                continue
            seen.add(ev)
        dist.extend(p for p in tags_prob_sum(rec, tags, vartype) if p is not None)
    dist.sort()
    return dist


def fdr_threshold_device(prob_dist: Sequence[float], alpha_ln: float, smart: bool = False, device: int = 0) -> Optional[float]:
    """The threshold search on the GPU (vlr_fdr_threshold, include/vlr.h; kernels in csrc/vlr_fdr.hip): sort, PEP prefix
    sums, boundary search.  `prob_dist`: the collected ln probabilities in ANY order, before the `smart` conversion."""
    import ctypes as C
    from . import engine
    L = engine.lib()
    L.vlr_fdr_threshold.restype = C.c_int
    L.vlr_fdr_threshold.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    a = np.ascontiguousarray(prob_dist, dtype=np.float64)
    thr, st = C.c_double(), C.c_int()
    rc = L.vlr_fdr_threshold(device, a.ctypes.data, len(a), int(smart), float(alpha_ln), C.byref(thr), C.byref(st))
    if rc != 0:
        raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
    return {0: None, 1: thr.value, 2: 0.0, 3: None}[st.value]


def fdr_threshold(prob_dist_desc: Sequence[float], alpha_ln: float, device: str = "cpu") -> Optional[float]:
    """fdr.rs:118-141 + bio::stats::bayesian::expected_fdr: threshold = probability of the last entry whose
    expected FDR (running mean of 1 - p over the descending list) is <= alpha.  Host restatement (numpy/torch on the CPU);
    `device="cuda"` routes to the HIP kernels behind vlr_fdr_threshold."""
    if str(device).startswith("cuda"):
        return fdr_threshold_device(prob_dist_desc, alpha_ln, smart=False, device=int(str(device).split(":")[1]) if ":" in str(device) else 0)
    import torch
    if len(prob_dist_desc) == 0:
        return None
    p = torch.tensor(prob_dist_desc, dtype=torch.float64, device=device)
    pep = torch.where(p < -0.693, torch.log1p(-torch.exp(p)), torch.log(-torch.expm1(p)))  # ln_one_minus_exp
    csum = torch.logcumsumexp(pep, dim=0)
    idx = torch.arange(1, len(p) + 1, dtype=torch.float64, device=device)
    fdr = torch.minimum(csum - torch.log(idx), torch.zeros_like(csum))
    if float(fdr[0]) > alpha_ln:
        return 0.0
    ok = fdr <= alpha_ln
    boundary = torch.ones_like(ok)
    boundary[1:] = pep[1:] != pep[:-1]  # do not let equal PEPs cross the boundary
    cand = torch.nonzero(ok & boundary).flatten()
    if len(cand) == 0:
        return None
    return float(p[int(cand[-1])])


def _relative_eq(a: float, b: float) -> bool:
    if a == b:
        return True
    if math.isinf(a) or math.isinf(b):
        return False
    d = abs(a - b)
    eps = np.finfo(np.float64).eps
    return d <= eps or d <= max(abs(a), abs(b)) * eps


def control_fdr(records: List[dict], events: Sequence[str], alpha: float, vartype=None, local: bool = False, smart: bool = False,
                smart_retain_artifacts: bool = False, header_tags: Optional[Sequence[str]] = None, device: str = "cpu") -> List[dict]:
    """fdr.rs:36-158; returns the kept records (alleles are not trimmed: single-ALT calls)."""
    tags = ["PROB_" + e.upper() for e in events]
    if header_tags is not None:
        tags = [t for t in tags if t in header_tags]
        if not tags:
            raise ValueError("invalid FDR control events")  # errors::Error::InvalidFDRControlEvents
    alpha_ln = math.log(alpha)
    threshold: Optional[float] = None
    if local:
        threshold = math.log1p(-alpha) if alpha < 1.0 else -math.inf
    elif alpha != 1.0:
        if smart:
            dist_tags = ["PROB_ABSENT"] + ([] if smart_retain_artifacts else ["PROB_ARTIFACT"])
        else:
            dist_tags = tags
        asc = collect_prob_dist(records, dist_tags, vartype)
        if str(device).startswith("cuda"):
            threshold = fdr_threshold_device(asc, alpha_ln, smart=smart, device=int(str(device).split(":")[1]) if ":" in str(device) else 0)
        else:
            desc = asc[::-1]
            if smart:
                desc = [(math.log1p(-math.exp(p)) if p < -0.693 else math.log(-math.expm1(p))) if p < 0 else -math.inf for p in desc]
            threshold = fdr_threshold(desc, alpha_ln, device=device)
    # filter_by_threshold (utils/mod.rs:288-374)
    ftags = list(tags)
    absent_tags = ["PROB_ABSENT"]
    if smart and smart_retain_artifacts:
        ftags.append("PROB_ARTIFACT")
    else:
        absent_tags.append("PROB_ARTIFACT")
    kept = []
    decisions: Dict[str, bool] = {}
    for rec in records:
        ev = rec["info"].get("EVENT") if isinstance(rec["info"].get("EVENT"), str) else None
        pe = tags_prob_sum(rec, ftags, vartype)
        pa = tags_prob_sum(rec, absent_tags, vartype) if smart else [None] * len(pe)
        keep_any = False
        for prob_events, prob_abs in zip(pe, pa):
            if ev is not None and ev in decisions:
                keep = decisions[ev]
            else:
                if smart:
                    p = None if prob_abs is None else ((math.log1p(-math.exp(prob_abs)) if prob_abs < -0.693 else math.log(-math.expm1(prob_abs))) if prob_abs < 0 else -math.inf)
                else:
                    p = prob_events
                if p is not None and threshold is not None:
                    keep = p > threshold or _relative_eq(p, threshold)
                elif p is not None and threshold is None:
                    keep = True
                else:
                    keep = False
                if smart:
                    keep = keep and (prob_events is not None and prob_events > LN_05)
                if ev is not None:
                    decisions[ev] = keep
            keep_any = keep_any or keep
        if keep_any:
            kept.append(rec)
    return kept


def filter_calls_native(in_path: str, out_path: str, events: Sequence[str], alpha: float, vartype=None, local: bool = False, smart: bool = False,
                        smart_retain_artifacts: bool = False, device: int = 0, threads: int = 0) -> Tuple[int, int]:
    """vlr_calls_filter_fdr (include/vlr.h): the whole command in the engine — calls BCF in, kept records out, threshold search on
    the device.  Returns (kept, total).  vartype = (kind, (lo, hi) | None) | None as control_fdr's."""
    import ctypes as C
    from . import engine
    L = engine.lib()
    L.vlr_calls_filter_fdr.restype = C.c_int
    L.vlr_calls_filter_fdr.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_double, C.c_uint32, C.c_char_p, C.c_int64, C.c_int64,
                                       C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    ev = (C.c_char_p * len(events))(*[e.encode() for e in events])
    mode = (1 if local else 0) | (2 if smart else 0) | (4 if smart_retain_artifacts else 0)
    kind, lo, hi = None, -1, -1
    if vartype is not None:
        kind, rng = vartype
        if rng is not None:
            lo, hi = int(rng[0]), int(min(rng[1], 1 << 62))
    kept, total = C.c_int64(), C.c_int64()
    rc = L.vlr_calls_filter_fdr(in_path.encode(), out_path.encode(), len(events), ev, float(alpha), mode, kind.encode() if kind else None, lo, hi,
                                int(device), int(threads), C.byref(kept), C.byref(total))
    if rc != 0:
        raise engine.EngineError(rc, (L.vlr_last_error() or b"").decode())
    return int(kept.value), int(total.value)
