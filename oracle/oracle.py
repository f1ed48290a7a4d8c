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
    src = os.path.join(_HERE, "vlr_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "vlr.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvlr_oracle.so"], stdout=subprocess.DEVNULL)
    return so


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
        _LIB = L
    return _LIB


def call(scenario, batch: PileupBatch, afd_capacity: int = 0, begin: int = 0, end: int = None, want_events=False):
    """Run the restated reference algorithm on loci [begin, end) of a host batch."""
    L = lib()
    end = batch.n_loci if end is None else end
    desc = scenario.desc()
    res = CallResults(batch.n_loci, scenario.n_out, batch.n_samples, afd_capacity)
    bs, rs = batch.as_struct(), res.as_struct()
    stats = Stats()
    ev = np.full((end - begin, 1 + 2 * len(scenario.event_names)), np.nan) if want_events else None
    rc = L.vlro_call_batch(C.byref(desc), C.byref(bs), C.byref(rs), begin, end,
                           ev.ctypes.data if ev is not None else None, C.byref(stats))
    if rc != 0:
        raise RuntimeError("oracle failed: %d" % rc)
    res.stats = {"n_lik_evals": stats.n_lik_evals, "n_obs_terms": stats.n_obs_terms, "n_joint": stats.n_joint}
    res.event_ln_posterior = ev
    return res
