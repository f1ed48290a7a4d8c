"""Device build of the shared decision arithmetic: bit-identical to the host build of include/vlr_detmath.h (that is the
premise of sharing it between kernel and oracle), and the kernel's own mantissa logarithm against libm."""
import ctypes as C

import numpy as np
import pytest

from test_detmath import SRC, _call  # noqa: F401  (host build of the same header)
from test_detmath import lib as hostlib  # noqa: F401

pytestmark = pytest.mark.gpu


def _dev(which, a, b=None):
    from varlociraptor_amd import engine
    L = engine.lib()
    L.vlr_selftest_math.restype = C.c_int
    L.vlr_selftest_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    out = np.empty(len(a))
    rc = L.vlr_selftest_math(0, which, a.ctypes.data, b.ctypes.data if b is not None else None, out.ctypes.data, len(a))
    assert rc == 0, L.vlr_last_error()
    return out


def test_device_detmath_is_bit_identical_to_host(hostlib):
    rng = np.random.default_rng(7)
    n = 400_000
    x = np.concatenate([rng.uniform(-745.0, 0.0, n), rng.uniform(-40.0, 2.0, n)])
    assert np.array_equal(_dev(0, x), _call(hostlib.t_exp, x))
    s = np.concatenate([10.0 ** rng.uniform(-300, 3, n), rng.uniform(0.0, 300.0, n)])
    assert np.array_equal(_dev(1, s), _call(hostlib.t_log1p, s))
    a, b = rng.uniform(1e-6, 1.0, 2 * n), rng.uniform(1e-6, 1.0, 2 * n)
    assert np.array_equal(_dev(2, a, b), _call(hostlib.t_log2r, a, b))
    v = np.concatenate([rng.integers(-20, 21, n).astype(float), rng.uniform(-20, 20, n)])
    assert np.array_equal(_dev(3, v), _call(hostlib.t_exp2, v))


def test_kernel_mantissa_log_is_within_one_ulp_of_libm():
    rng = np.random.default_rng(8)
    m = np.concatenate([rng.uniform(0.5, 1.0, 1_000_000), [0.5, np.nextafter(1.0, 0.0), np.sqrt(0.5), np.nextafter(np.sqrt(0.5), 0.0)]])
    got = _dev(4, m)
    ref = np.log(m)
    assert (np.abs(got - ref) <= 1.0 * np.spacing(np.abs(ref)) + 1e-300).all()
