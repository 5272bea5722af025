"""The edge solver's own binary64 logarithm (admm_amd/csrc/edge_log.h; Tree.cpp:120-133 take the
logarithm of every marginal).  On the CPU: its host build against the long double logarithm
(below 0.76 units in the last place, exact at 1, NaN and infinity passed through) and against the
committed bits.  On the GPU: the device gives the same bits for the same arguments."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)


@pytest.fixture(scope="module")
def host_program(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("edge_log") / "check_edge_log")
    subprocess.check_call(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_edge_log.cpp")])
    return exe


def test_host_build_accuracy(host_program):
    out = subprocess.run([host_program, "1000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "log(0x1p+0) = 0x0p+0" in out.stdout
    assert "log(inf) = inf" in out.stdout and "log(nan) = nan" in out.stdout


def test_host_build_gives_the_committed_bits(host_program):
    import make_edge_log as M
    gold = np.load(os.path.join(GOLD, "edge_log.npz"))
    assert np.array_equal(M.arguments(), gold["x"])
    y = M.host_values(gold["x"], host_program)
    assert np.array_equal(y.view(np.uint64), gold["y_bits"])
    # against the C library's log as numpy calls it: never more than one unit apart, and the
    # value the solver stores (rounded to binary32) the same but for a handful of arguments
    ref = np.log(gold["x"])
    fin = np.isfinite(ref) & (ref != 0)
    assert np.all(np.abs(y[fin] - ref[fin]) <= np.spacing(np.abs(ref[fin])))
    assert np.count_nonzero(y.astype(np.float32) != ref.astype(np.float32)) <= 2


def test_division_by_the_rounded_reciprocal_is_the_division(tmp_path):
    """The solve's mu1 * ex / acc (Tree.cpp:271) as a binary64 product with 1 / acc and one
    residual step: bit for bit the binary32 division, on random pairs, on pairs built next to
    rounding boundaries, around the overflow and the underflow thresholds and on exact ties
    between two subnormal results."""
    exe = str(tmp_path / "check_recip_div")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_recip_div.cpp")])
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "20000000 pairs: 0 results differ" in out.stdout and "DIFFERENT" not in out.stdout


@pytest.mark.gpu
def test_device_gives_the_host_builds_bits():
    import ctypes as C
    from admm_amd._lib import lib
    gold = np.load(os.path.join(GOLD, "edge_log.npz"))
    x = np.ascontiguousarray(gold["x"])
    x = np.concatenate([x, [np.nan]])
    y = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    assert lib.admm_edge_log_values(0, x.ctypes.data_as(dp), y.ctypes.data_as(dp), x.size) == 0
    assert np.array_equal(y[:-1].view(np.uint64), gold["y_bits"])
    assert np.isnan(y[-1])
