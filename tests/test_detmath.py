"""include/vlr_detmath.h is compiled into BOTH the kernel and the oracle (decision sums of the bias gating, l2fc
predicates), so a defect there would be common-mode and invisible to every parity test.  Here the
functions are checked against libm on a million random arguments each (CPU build of the same header; the header uses only
IEEE +,-,*,/ and fma in a fixed order, so the gfx950 build computes the same bits)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "vlr_detmath.h"
extern "C" {
void t_exp(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = vlr_det::det_exp(x[i]); }
void t_log1p(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = vlr_det::det_log1p_pos(x[i]); }
void t_log2r(const double* a, const double* b, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = vlr_det::det_log2_ratio(a[i], b[i]); }
void t_exp2(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = vlr_det::det_exp2(x[i]); }
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("detmath")
    src = d / "t.cpp"
    src.write_text(SRC)
    so = d / "t.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(so)])
    return C.CDLL(str(so))


def _call(f, *arrs):
    n = len(arrs[0])
    out = np.empty(n)
    f(*[a.ctypes.data_as(C.c_void_p) for a in arrs], out.ctypes.data_as(C.c_void_p), C.c_long(n))
    return out


def _ulps(got, ref):
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.abs(got - ref) / np.spacing(np.abs(ref))


N = 1_000_000


def test_det_exp_matches_libm(lib):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-745.0, 0.0, N // 2), rng.uniform(-40.0, 0.0, N // 2), [0.0, -1e-300, -700.0]])
    got = _call(lib.t_exp, x)
    ref = np.exp(x)
    ok = ref > 1e-300  # gradual underflow is outside the decision range (sums of exp(pm - max) with pm - max in [-745, 0])
    assert _ulps(got[ok], ref[ok]).max() <= 2.0


def test_det_log1p_matches_libm(lib):
    rng = np.random.default_rng(2)
    x = np.concatenate([10.0 ** rng.uniform(-300, 3, N // 2), rng.uniform(0.0, 300.0, N // 2), [0.0, 1.0, 1e-17]])
    got = _call(lib.t_log1p, x)
    ref = np.log1p(x)
    # the function forms u = 1 + s first, so its error is ABSOLUTE (one rounding of u, 2^-52) — which is what its callers
    # need: the result is added to the maximum of a ln_sum_exp (strand_bias.rs:80-109) and exponentiated
    err = np.abs(got - ref)
    assert (err <= np.maximum(2.0 * np.spacing(ref), 2.3e-16)).all()
    assert got[x == 0.0][0] == 0.0


def test_det_log2_ratio_and_exp2(lib):
    rng = np.random.default_rng(3)
    a, b = rng.uniform(1e-6, 1.0, N), rng.uniform(1e-6, 1.0, N)
    got = _call(lib.t_log2r, a, b)
    ref = np.log2(a / b)
    err = np.abs(got - ref)
    assert (err <= 4 * np.spacing(np.maximum(np.abs(ref), 1.0))).all()
    # exact on power-of-two ratios: the reason the function exists (l2fc end points, log2_fold_change.rs:17-93)
    k = rng.integers(-20, 21, 10000)
    a2 = rng.uniform(1e-3, 1.0, 10000)
    assert np.array_equal(_call(lib.t_log2r, a2 * 2.0 ** k, a2), k.astype(float))
    v = np.concatenate([k.astype(float), rng.uniform(-20, 20, N // 10)])
    got2 = _call(lib.t_exp2, v)
    assert np.array_equal(got2[:10000], 2.0 ** k)
    assert _ulps(got2, np.exp2(v)).max() <= 2.0
