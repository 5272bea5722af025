"""The edge solver's exponential (the device library's algorithm without its per-call range
checks: the argument is clamped, a NaN argument is flagged) against the device library's exp
on the same binary32 arguments: the same binary64 bits everywhere, the ends of the range and
the infinities included; NaN arguments are flagged."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _values(x):
    from admm_amd._lib import lib
    x = np.ascontiguousarray(x, dtype=np.float32)
    own, ref = np.empty(x.size), np.empty(x.size)
    nan = np.empty(x.size, dtype=np.uint8)
    dp = C.POINTER(C.c_double)
    rc = lib.admm_edge_exp_values(0, x.ctypes.data_as(C.POINTER(C.c_float)), own.ctypes.data_as(dp), ref.ctypes.data_as(dp),
                                  nan.ctypes.data_as(C.POINTER(C.c_ubyte)), x.size)
    assert rc == 0
    return own, ref, nan


def test_edge_exp_equals_the_device_library_bit_for_bit():
    rng = np.random.default_rng(5)
    x = np.concatenate([
        rng.uniform(-120.0, 100.0, 400000), rng.uniform(-1.0, 1.0, 100000), rng.uniform(-1200.0, 1200.0, 100000),
        rng.standard_normal(100000) * 1e-6,
        [0.0, -0.0, 88.0, 88.72284, 89.0, -87.33655, -103.97, -104.0, 709.78, 709.79, 710.0, -745.13, -745.2, 1024.0, 1025.0, 1026.0,
         -1075.0, -1076.0, -1077.0, 1e10, -1e10, 3.4e38, -3.4e38, np.inf, -np.inf, 1e-45, -1e-45],
    ]).astype(np.float32)
    own, ref, nan = _values(x)
    assert not nan.any()
    assert np.array_equal(own.view(np.uint64), ref.view(np.uint64))
    # and rounded to binary32, where the solver stores it
    with np.errstate(over="ignore"):
        assert np.array_equal(own.astype(np.float32).view(np.uint32), ref.astype(np.float32).view(np.uint32))
    assert own[x == np.inf][0] == np.inf and own[x == -np.inf][0] == 0.0


def test_edge_exp_flags_nan_arguments():
    x = np.array([1.0, np.nan, -3.0, np.nan], dtype=np.float32)
    own, ref, nan = _values(x)
    assert nan.tolist() == [0, 1, 0, 1]
    assert np.isnan(ref[1]) and np.isnan(ref[3])
    assert own[0] == ref[0] and own[2] == ref[2]
