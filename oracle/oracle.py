"""ctypes binding of the CPU oracle (oracle/libvlr_oracle.so).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from the
product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from varlociraptor_amd import abi
from varlociraptor_amd.batch import CallResults, PileupBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Stats(C.Structure):
    _fields_ = [("n_lik_evals", C.c_uint64), ("n_obs_terms", C.c_uint64), ("n_joint", C.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "libvlr_oracle.so")
    so2 = os.path.join(_HERE, "libvlr_oracle_tuned.so")
    src = os.path.join(_HERE, "vlr_oracle.cpp")
    src2 = os.path.join(_HERE, "vlr_realign_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "vlr.h")
    newest = max(os.path.getmtime(src), os.path.getmtime(src2), os.path.getmtime(hdr))
    if force or not os.path.exists(so) or not os.path.exists(so2) or min(os.path.getmtime(so), os.path.getmtime(so2)) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return so


_LIB_TUNED = None


def lib_tuned():
    """The same source built for AVX2/FMA hosts (-O3 -march=x86-64-v3): only vlro_call_batch_tuned is used from it."""
    global _LIB_TUNED
    if _LIB_TUNED is None:
        so = os.path.join(_HERE, "libvlr_oracle_tuned.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.vlro_call_batch_tuned.restype = C.c_int
        L.vlro_call_batch_tuned.argtypes = [C.POINTER(abi.ScenarioDesc), C.POINTER(abi.Batch), C.POINTER(abi.Results),
                                            C.c_int64, C.c_int64, C.c_void_p, C.POINTER(Stats)]
        _LIB_TUNED = L
    return _LIB_TUNED


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvlr_oracle.so")
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        L.vlro_call_batch.restype = C.c_int
        L.vlro_call_batch.argtypes = [C.POINTER(abi.ScenarioDesc), C.POINTER(abi.Batch), C.POINTER(abi.Results),
                                      C.c_int64, C.c_int64, C.c_void_p, C.POINTER(Stats)]
        for name, n in (("vlro_ln_one_minus_exp", 1), ("vlro_ln_add_exp", 2)):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_double] * n
        L.vlro_ln_sum_exp.restype = C.c_double
        L.vlro_ln_sum_exp.argtypes = [C.POINTER(C.c_double), C.c_int]
        for name in ("vlro_observable_min", "vlro_observable_max"):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        L.vlro_adaptive_gauss.restype = C.c_double
        L.vlro_adaptive_gauss.argtypes = [C.c_double] * 5 + [C.POINTER(C.c_int)]
        L.vlro_prior.restype = C.c_double
        L.vlro_prior.argtypes = [C.POINTER(abi.ScenarioDesc), C.POINTER(C.c_double), C.c_int]
        L.vlro_lik_obs_single.restype = C.c_double
        L.vlro_lik_obs_single.argtypes = [C.POINTER(abi.Batch), C.c_int64, C.c_double]
        L.vlro_lik_obs_contaminated.restype = C.c_double
        L.vlro_lik_obs_contaminated.argtypes = [C.POINTER(abi.Batch), C.c_int64, C.c_double, C.c_double, C.c_double]
        L.vlro_pileup_lik_single.restype = C.c_double
        L.vlro_pileup_lik_single.argtypes = [C.POINTER(abi.Batch), C.c_int64, C.c_int64, C.c_double]
        L.vlro_pileup_lik_contaminated.restype = C.c_double
        L.vlro_pileup_lik_contaminated.argtypes = [C.POINTER(abi.Batch), C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double]
        L.vlro_bias_prob_ref_none.restype = C.c_double
        L.vlro_bias_prob_ref_none.argtypes = [C.POINTER(abi.Batch), C.c_int64]
        L.vlro_pairhmm_prob_related.restype = C.c_double
        L.vlro_pairhmm_prob_related.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.vlro_edit_distance.restype = C.c_int
        L.vlro_edit_distance.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vlro_normalize_support.restype = None
        L.vlro_normalize_support.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


def pairhmm_prob_related(allele: bytes, read: bytes, qual, gap, max_edit_dist: int = -1) -> float:
    """Restated bio PairHMM::prob_related over the reference's emission model (oracle/vlr_realign_oracle.cpp)."""
    L = lib()
    x = np.frombuffer(bytes(allele) or b"\0", np.uint8)
    y = np.frombuffer(bytes(read) or b"\0", np.uint8)
    q = np.asarray(bytearray(qual) or b"\0", np.uint8)
    g = (C.c_double * 4)(*gap)
    return float(L.vlro_pairhmm_prob_related(x.ctypes.data, len(allele), y.ctypes.data, q.ctypes.data, len(read), g, int(max_edit_dist)))


def pairhmm_prob_related_variant(allele: bytes, read: bytes, qual, gap, max_edit_dist: int = -1, crate_behaviours: int = 7) -> float:
    """The recursion with the crate behaviours of the oracle's header switched on (bit 0 approximate 3-way sum, bit 1 stale gap
    states of skipped band cells, bit 2 doubled start mass of the first column)."""
    L = lib()
    L.vlro_pairhmm_prob_related_variant.restype = C.c_double
    L.vlro_pairhmm_prob_related_variant.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int]
    x = np.frombuffer(bytes(allele) or b"\0", np.uint8)
    y = np.frombuffer(bytes(read) or b"\0", np.uint8)
    q = np.asarray(bytearray(qual) or b"\0", np.uint8)
    g = (C.c_double * 4)(*gap)
    return float(L.vlro_pairhmm_prob_related_variant(x.ctypes.data, len(allele), y.ctypes.data, q.ctypes.data, len(read), g, int(max_edit_dist), int(crate_behaviours)))


def pathhmm_best(allele: bytes, read: bytes, qual, gap) -> float:
    """`fast` realignment mode: best path probability over the minimal-edit-distance alignments (vlro_pathhmm_best)."""
    L = lib()
    L.vlro_pathhmm_best.restype = C.c_double
    L.vlro_pathhmm_best.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    x = np.frombuffer(bytes(allele) or b"\0", np.uint8)
    y = np.frombuffer(bytes(read) or b"\0", np.uint8)
    q = np.asarray(bytearray(qual) or b"\0", np.uint8)
    g = (C.c_double * 4)(*gap)
    return float(L.vlro_pathhmm_best(x.ctypes.data, len(allele), y.ctypes.data, q.ctypes.data, len(read), g))


def pathhmm_fixed_traceback(allele: bytes, read: bytes, qual, gap):
    """(ln path probability of the alignment a diagonal-first traceback picks, number of co-optimal alignments at its end position):
    measurement aid for the `fast` mode (vlro_pathhmm_fixed_traceback)."""
    L = lib()
    L.vlro_pathhmm_fixed_traceback.restype = C.c_double
    L.vlro_pathhmm_fixed_traceback.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    x = np.frombuffer(bytes(allele) or b"\0", np.uint8)
    y = np.frombuffer(bytes(read) or b"\0", np.uint8)
    q = np.asarray(bytearray(qual) or b"\0", np.uint8)
    g = (C.c_double * 4)(*[float(v) for v in gap])
    n = C.c_double()
    p = float(L.vlro_pathhmm_fixed_traceback(x.ctypes.data, len(allele), y.ctypes.data, q.ctypes.data, len(read), g, C.byref(n)))
    return p, float(n.value)


def homopoly_prob_related(allele: bytes, read: bytes, qual, gap, hop, max_edit_dist: int = -1) -> float:
    """`homopolymer` realignment mode: restated bio HomopolyPairHMM::prob_related (vlro_homopoly_prob_related); hop = 16 ln values."""
    L = lib()
    L.vlro_homopoly_prob_related.restype = C.c_double
    L.vlro_homopoly_prob_related.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    x = np.frombuffer(bytes(allele) or b"\0", np.uint8)
    y = np.frombuffer(bytes(read) or b"\0", np.uint8)
    q = np.asarray(bytearray(qual) or b"\0", np.uint8)
    g = (C.c_double * 4)(*[float(v) for v in gap])
    h = (C.c_double * 16)(*[float(v) for v in hop])
    return float(L.vlro_homopoly_prob_related(x.ctypes.data, len(allele), y.ctypes.data, q.ctypes.data, len(read), g, h, int(max_edit_dist)))


def homopoly_batch(batch, gap, hop):
    g = [gap.prob_insertion_artifact, gap.prob_deletion_artifact, gap.prob_insertion_extend_artifact, gap.prob_deletion_extend_artifact]
    h = hop.as_list()
    return np.array([homopoly_prob_related(batch.x[k], batch.y[k], batch.q[k], g, h, batch.band[k]) for k in range(len(batch))])


def edit_distance(allele: bytes, read: bytes):
    """(dist, end, n_hits) of the semiglobal edit distance of `read` in `allele` (oracle/vlr_realign_oracle.cpp)."""
    L = lib()
    x = np.frombuffer(bytes(allele) or b"\0", np.uint8)
    y = np.frombuffer(bytes(read) or b"\0", np.uint8)
    e, n = C.c_int(), C.c_int()
    d = L.vlro_edit_distance(x.ctypes.data, len(allele), y.ctypes.data, len(read), C.byref(e), C.byref(n))
    return int(d), int(e.value), int(n.value)


def pairhmm_batch(batch, gap, threads=1):
    """All pairs of a varlociraptor_amd.realign.PairBatch through the oracle."""
    g = [gap.prob_insertion_artifact, gap.prob_deletion_artifact, gap.prob_insertion_extend_artifact, gap.prob_deletion_extend_artifact]
    idx = range(len(batch))
    f = lambda k: pairhmm_prob_related(batch.x[k], batch.y[k], batch.q[k], g, batch.band[k])
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as ex:
            return np.array(list(ex.map(f, idx)))
    return np.array([f(k) for k in idx])


def normalize_support(prob_ref: float, prob_alt: float):
    r, a = C.c_double(prob_ref), C.c_double(prob_alt)
    lib().vlro_normalize_support(C.byref(r), C.byref(a))
    return r.value, a.value


def call(scenario, batch: PileupBatch, afd_capacity: int = 0, begin: int = 0, end: int = None, want_events=False, tuned=False):
    """Run the restated reference algorithm on loci [begin, end) of a host batch.  tuned=True: the tuned CPU baseline (affine
    product form of the pileup likelihood, libvlr_oracle_tuned.so built with -march=x86-64-v3)."""
    L = lib()
    end = batch.n_loci if end is None else end
    desc = scenario.desc()
    res = CallResults(batch.n_loci, scenario.n_out, batch.n_samples, afd_capacity)
    bs, rs = batch.as_struct(), res.as_struct()
    stats = Stats()
    ev = np.full((end - begin, 1 + 2 * len(scenario.event_names)), np.nan) if want_events else None
    fn = lib_tuned().vlro_call_batch_tuned if tuned else L.vlro_call_batch
    rc = fn(C.byref(desc), C.byref(bs), C.byref(rs), begin, end,
                           ev.ctypes.data if ev is not None else None, C.byref(stats))
    if rc != 0:
        raise RuntimeError("oracle failed: %d" % rc)
    res.stats = {"n_lik_evals": stats.n_lik_evals, "n_obs_terms": stats.n_obs_terms, "n_joint": stats.n_joint}
    res.event_ln_posterior = ev
    return res
