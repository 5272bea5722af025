"""CPU tests of the host side: the C ABI library loads and exports every symbol
include/admm_hip.h declares; partition arithmetic; error behaviour without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "admm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(admm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from admm_amd import _lib
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_lib.lib, n), "libadmm_hip.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names, "admm_amd/_lib.py and include/admm_hip.h disagree"


def test_desc_layout_matches_header():
    from admm_amd._lib import AdmmDesc
    # 12 x 4-byte fields, one pointer (8-byte aligned), two ints
    assert AdmmDesc.edge_list.offset == 48
    assert C.sizeof(AdmmDesc) == 64


def test_partition_matches_reference_formula():
    import admm_amd as A
    for n in (1, 7, 64, 2048, 2051):
        for G in (1, 2, 3, 8):
            total, prev_end = 0, 0
            for r in range(G):
                # Grid.cpp:377-390
                extra = n % G
                if r < extra:
                    size = n // G + 1
                    off = r * size
                else:
                    size = n // G
                    off = extra * ((n + G - 1) // G) + (r - extra) * size
                assert A.partition(n, G, r) == (off, size)
                assert off == prev_end
                prev_end = off + size
                total += size
            assert total == n


def test_generators_match_oracle():
    import admm_amd as A
    from oracle import oracle as O
    a, sa = A.lcg_fill(1234, 1000)
    b, sb = O.lcg_fill(1234, 1000)
    assert np.array_equal(a, b) and sa == sb
    assert a.min() >= 0.0 and a.max() < 1.0
    for (R, Cc) in ((1, 1), (1, 5), (4, 3), (7, 7)):
        assert np.array_equal(A.grid_edges(R, Cc), O.grid_edges(R, Cc))
    # Grid::Generate's order on the reference's own toy size (Grid.cpp:581-615, 4 x 3)
    assert A.grid_edges(4, 3)[:10].tolist() == [0, 1, 1, 2, 0, 3, 1, 4, 2, 5]
    for kind in ("potts", "trunc_linear"):
        assert np.array_equal(A.pairwise_table(kind, 8), O.pairwise_table(kind, 8))


def test_create_rejects_bad_arguments_and_reports_why():
    from admm_amd._lib import AdmmDesc, lib
    h = C.c_void_p()
    d = AdmmDesc()
    d.rows, d.cols, d.k, d.batch, d.solver = 4, 4, 1, 1, 1
    d.penalty, d.penalty_prime, d.rho, d.nranks = 0.5, 1.0, 1.0, 1
    assert lib.admm_create(C.byref(d), C.byref(h)) == -1
    assert b"k must be" in lib.admm_last_error(None)
    d.k = 65  # the chain solver holds a label vector across one wavefront's lanes
    assert lib.admm_create(C.byref(d), C.byref(h)) == -1
    assert b"k must be <= 64" in lib.admm_last_error(None)
    d.k, d.solver = 1025, 0  # the edge solver goes on to 1024 labels
    assert lib.admm_create(C.byref(d), C.byref(h)) == -1
    assert b"k must be <= 1024" in lib.admm_last_error(None)
    d.solver = 1
    d.k, d.solver = 4, 7
    assert lib.admm_create(C.byref(d), C.byref(h)) == -1
    assert b"solver" in lib.admm_last_error(None)
    d.solver, d.nranks, d.rank, d.batch = 0, 2, 0, 2
    assert lib.admm_create(C.byref(d), C.byref(h)) == -1
    assert b"edge solver" in lib.admm_last_error(None)
    d.solver, d.batch = 1, 3
    assert lib.admm_create(C.byref(d), C.byref(h)) == -1
    assert b"batch" in lib.admm_last_error(None)
    # a column chain whose half would span 4 GiB or more is refused, not wrapped
    d.batch, d.nranks, d.rows, d.cols, d.k = 1, 1, 70000, 1000, 32
    assert lib.admm_create(C.byref(d), C.byref(h)) == -1
    assert b"too large" in lib.admm_last_error(None)


def test_no_cpu_fallback():
    """Without a HIP device the product path fails loudly instead of computing elsewhere."""
    import admm_amd as A
    from admm_amd._lib import AdmmDesc, lib
    d = AdmmDesc()
    d.rows, d.cols, d.k, d.batch, d.solver = 4, 4, 4, 1, 1
    d.penalty, d.penalty_prime, d.rho, d.nranks = 0.5, 1.0, 1.0, 1
    h = C.c_void_p()
    rc = lib.admm_create(C.byref(d), C.byref(h))
    if rc == 0:
        lib.admm_destroy(h)
        pytest.skip("a HIP device is present")
    assert rc == -3
    assert b"HIP device" in lib.admm_last_error(None)
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetGlobalInfo(4, 4, 4)
    g.Generate()
    with pytest.raises(A.AdmmError):
        g.InitOptimization()


def test_exchange_plan_round_trip():
    """pack on every rank + unpack on every rank reproduces the global label arrays."""
    from admm_amd import strips as P
    rng = np.random.default_rng(3)
    for (R, Cc, G) in ((8, 8, 2), (7, 10, 3), (5, 3, 4), (16, 9, 8)):
        xrow = rng.integers(0, 9, size=(R, Cc)).astype(np.int32)
        xcol = rng.integers(0, 9, size=(R, Cc)).astype(np.int32)
        st = P.strips(R, Cc, G)
        sends = []
        for g, (r0, nr, c0, nc) in enumerate(st):
            sends.append(P.pack(xrow[r0:r0 + nr, :], np.ascontiguousarray(xcol[:, c0:c0 + nc].T), R, Cc, G))
        for g, (r0, nr, c0, nc) in enumerate(st):
            ra = [sends[h][0][g] for h in range(G)]
            rb = [sends[h][1][g] for h in range(G)]
            xrow_c, xcolT_r = P.unpack(ra, rb, R, Cc, G, g)
            assert np.array_equal(xrow_c, xrow[:, c0:c0 + nc])
            assert np.array_equal(xcolT_r, xcol[r0:r0 + nr, :].T)
