"""Fixed GPU regression cases: one that a random sweep once failed on, the headline problem
cut into eight strips at full size, and call orders no other test walks through."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _strips(R, C, k, un, W, G, exchange):
    import admm_amd as A
    gs = []
    for r in range(G):
        g = A.Grid(rank=r, nprocs=G)
        g.exchange = exchange
        g.SetPartitioner(A.ROW)
        g.SetPenalties(0.5, 1.0)
        g.SetGlobalInfo(R, C, k)
        g.Load(un, W)
        g.InitOptimization()
        gs.append(g)
    return gs


def _iterate(gs):
    import admm_amd as A
    for g in gs:
        g.Compute()
    if len(gs) > 1:
        A.communicate_group(gs)
    for g in gs:
        g.Update()


def test_random_sweep_case_53x14_k48_seven_strips_dense_exchange(monkeypatch):
    """Case 300 of the random parity sweep with seed 77 failed once ("labels, iteration 1")
    and never again: 53 x 14, k = 48, seven strips, dense exchange, Potts, unaries from
    LCG seed 912101, right after hundreds of other contexts had used and freed the same
    memory.  What that record points at is set-up, not the sweep: at the time the
    create-time memsets and uploads ran on the null stream while the contexts' own streams
    do not synchronise with it, so a first sweep could start on label arrays that were not
    yet -1.  They now run on the context's stream.  Here every allocation is pre-filled
    with a byte that is not 0xFF, so that label arrays nothing initialised cannot read as -1."""
    from oracle import oracle as O
    monkeypatch.setenv("ADMM_POISON", "165")
    R, C, k, G = 53, 14, 48, 7
    un, W = O.synth_problem(R, C, k, 912101, "potts")
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    gs = _strips(R, C, k, un, W, G, exchange=1)
    for it in range(4):
        o.compute(); o.update()
        _iterate(gs)
        lab = np.concatenate([g.Labels().reshape(-1, C) for g in gs if g.StripInfo()[1] > 0], axis=0)
        assert np.array_equal(lab, o.labels()), "labels, iteration %d" % (it + 1)
    lam = np.concatenate([g.State(10).reshape(-1, C, k) for g in gs if g.StripInfo()[1] > 0], axis=0)
    assert np.array_equal(lam, o.lambda_row)
    lamc = np.concatenate([g.State(11).reshape(R, -1, k) for g in gs if g.StripInfo()[3] > 0], axis=1)
    assert np.array_equal(lamc, o.lambda_col)
    for g in gs:
        g.close()


@pytest.mark.parametrize("exchange", [0, 1], ids=["labels", "dense"])
def test_config_2048x2048_k32_eight_strips(exchange):
    """The headline configuration as stated: 2048 x 2048, k = 32, cut into 8 row / column
    strips (all eight contexts on this one GPU, exchanged in process), against the oracle."""
    from oracle import oracle as O
    R = C = 2048
    k, G = 32, 8
    iters = 20 if exchange == 0 else 3  # labels: past the first scan of the int8 counts (sweep 16)
    O.set_threads(min(16, os.cpu_count() or 1))
    un, W = O.synth_problem(R, C, k, 1234, "potts")
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    gs = _strips(R, C, k, un, W, G, exchange)
    for _ in range(iters):
        o.compute(); o.update()
        _iterate(gs)
    prim = dual = lin = 0.0
    for g in gs:
        r0, nr, c0, nc = g.StripInfo()
        assert (nr, nc) == (256, 256)
        assert np.array_equal(g.Labels(), o.labels()[r0:r0 + nr])
        assert np.array_equal(g.State(13).reshape(nr, C), o.xrow[r0:r0 + nr])
        assert np.array_equal(g.State(14).reshape(R, nc), o.xcol[:, c0:c0 + nc])
        assert np.array_equal(g.State(10).reshape(nr, C, k), o.lambda_row[r0:r0 + nr])
        assert np.array_equal(g.State(11).reshape(R, nc, k), o.lambda_col[:, c0:c0 + nc])
        p, d = g.Objective()
        prim += p; dual += d; lin += g.ComputeLinearObj()
    po, do, lo = o.objective()
    assert abs(prim - po) <= 1e-9 * abs(po) and abs(dual - do) <= 1e-9 * abs(do) and abs(lin - lo) <= 1e-9 * abs(lo)
    for g in gs:
        g.close()


@pytest.mark.parametrize("rho", [1.0, 2.0])
def test_new_pairwise_table_and_second_init_keep_the_unary_shares(rho):
    """upload_unaries, upload_pairwise, init, then a new table and init again: the per-copy
    shares of the unaries are worked out in place by the first init and must not be divided
    a second time."""
    import ctypes as C
    import admm_amd as A
    from admm_amd._lib import lib
    from oracle import oracle as O
    R, Cc, k = 24, 37, 16
    un, W1 = O.synth_problem(R, Cc, k, 3, "potts")
    W2 = O.pairwise_table("trunc_linear", k)
    fp = C.POINTER(C.c_float)

    def fresh(W):
        g = A.Grid()
        g.SetPartitioner(A.ROW)
        g.SetPenalties(0.5, 1.0)
        g.SetRhoEdge(rho)
        g.SetGlobalInfo(R, Cc, k)
        g.Load(un, W)
        g.InitOptimization()
        return g

    g = fresh(W1)
    for _ in range(3):
        g.Compute(); g.Communicate(); g.Update()
    w2 = np.ascontiguousarray(W2, dtype=np.float32)
    assert lib.admm_upload_pairwise(g._h, w2.ctypes.data_as(fp), 0) == 0
    assert lib.admm_init(g._h) == 0
    ref = fresh(W2)
    assert np.array_equal(g.State(12), ref.State(12)), "unary shares changed by the second init"
    assert (g.State(13) == -1).all() and not g.State(10).any()
    for _ in range(4):
        for x in (g, ref):
            x.Compute(); x.Communicate(); x.Update()
    assert np.array_equal(g.Labels(), ref.Labels())
    assert np.array_equal(g.State(10), ref.State(10))
    o = O.ChainOracle(R, Cc, k, un, W2, rho=rho)
    o.init()
    for _ in range(4):
        o.compute(); o.update()
    assert np.array_equal(g.Labels(), o.labels())
    g.close(); ref.close()


@pytest.mark.parametrize("G", [1, 3, 8])
def test_strips_that_read_their_own_share_of_a_file(tmp_path, G, monkeypatch):
    """Every strip of a grid file reads only its rows and its columns (as a rank of the
    reference reads only its hyperslab, Grid.cpp:446-454, 510-523), in small blocks; the
    result equals strips that were handed the whole array."""
    import admm_amd as A
    from admm_amd import graphfile
    from oracle import oracle as O
    R, C, k = 45, 52, 8
    un, W = O.synth_problem(R, C, k, 99, "potts")
    path = tmp_path / "grid.nc"
    graphfile.write(path, R * C, O.grid_edges(R, C), -un, -W, rows=R, cols=C)
    monkeypatch.setattr(A.Grid, "UPLOAD_BLOCK_BYTES", 4 * k * C * 7)  # seven grid rows per block
    whole = _strips(R, C, k, un, W, G, 0)
    gs = []
    for r in range(G):
        g = A.Grid(rank=r, nprocs=G)
        g.SetPartitioner(A.ROW)
        g.SetPenalties(0.5, 1.0)
        g.SetGlobalInfo(str(path))
        g.Load(str(path))
        if G > 1:
            assert g._unary is None, "a strip of several must not hold the whole array"
        g.InitOptimization()
        gs.append(g)
    for a, b in zip(gs, whole):
        assert np.array_equal(a.State(12), b.State(12))
    for _ in range(4):
        _iterate(gs); _iterate(whole)
    for a, b in zip(gs, whole):
        assert np.array_equal(a.Labels(), b.Labels())
        assert np.array_equal(a.State(10), b.State(10)) and np.array_equal(a.State(11), b.State(11))
    for g in gs + whole:
        g.close()


def test_share_upload_call_order():
    import ctypes as C
    from admm_amd._lib import AdmmDesc, lib
    d = AdmmDesc()
    d.rows, d.cols, d.k, d.batch, d.solver = 6, 5, 4, 1, 1
    d.penalty, d.penalty_prime, d.rho, d.nranks, d.rank = 0.5, 1.0, 1.0, 2, 1
    h = C.c_void_p()
    assert lib.admm_create(C.byref(d), C.byref(h)) == 0
    fp = C.POINTER(C.c_float)
    rows = np.zeros(3 * 5 * 4, dtype=np.float32)
    cols = np.zeros(6 * 2 * 4, dtype=np.float32)
    w = np.zeros(16, dtype=np.float32)
    assert lib.admm_upload_pairwise(h, w.ctypes.data_as(fp), 0) == 0
    assert lib.admm_upload_unaries_rows(h, 0, 3, rows.ctypes.data_as(fp)) == -1  # rank 1 owns rows 3..5
    assert b"outside" in lib.admm_last_error(h)
    assert lib.admm_upload_unaries_rows(h, 3, 3, rows.ctypes.data_as(fp)) == 0
    assert lib.admm_init(h) == -2  # the column share is still missing
    assert lib.admm_upload_unaries_cols(h, 0, 4, cols.ctypes.data_as(fp)) == 0
    assert lib.admm_init(h) == -2  # rows 4 and 5 of it
    assert lib.admm_upload_unaries_cols(h, 4, 2, cols.ctypes.data_as(fp)) == 0
    assert lib.admm_init(h) == 0
    assert lib.admm_compute(h) == 0
    # a share after a solve starts a new upload
    assert lib.admm_upload_unaries_rows(h, 3, 3, rows.ctypes.data_as(fp)) == 0
    assert lib.admm_compute(h) == -2 and lib.admm_init(h) == -2
    lib.admm_destroy(h)


def test_consensus_of_an_isolated_vertex_is_uniform():
    """A user graph may hold a vertex no edge touches: its consensus stays the uniform 1/k
    of Tree.cpp:24-33 instead of 0/0."""
    import admm_amd as A
    k = 3
    g = A.Grid()
    g.SetPartitioner(A.EDGE)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(1, 1, k)
    g.SetGraph(4, np.array([0, 1, 1, 2], dtype=np.int32))  # vertex 3 is isolated
    rng = np.random.default_rng(0)
    g.Load(rng.random((4, k), dtype=np.float32), rng.random((2, k * k), dtype=np.float32), per_edge=True)
    g.InitOptimization()
    for _ in range(3):
        g.Compute(); g.Communicate(); g.Update()
    m = g.Consensus()
    assert np.isfinite(m).all()
    assert np.array_equal(m[3], np.full(k, np.float32(1.0 / np.float32(k)), dtype=np.float32))
    g.close()


def test_init_optimization_checks_sizes_and_replaces_the_context():
    import admm_amd as A
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(8, 8, 4)
    g.Load(np.zeros((8 * 8 - 1, 4), dtype=np.float32), np.zeros((4, 4), dtype=np.float32))
    with pytest.raises(ValueError):
        g.InitOptimization()
    g.Load(np.zeros((8 * 8, 4), dtype=np.float32), np.zeros((3, 3), dtype=np.float32))
    with pytest.raises(ValueError):
        g.InitOptimization()
    g.Load(np.zeros((8 * 8, 4), dtype=np.float32), np.zeros((4, 4), dtype=np.float32))
    g.InitOptimization()
    first = g._h.value
    g.InitOptimization()  # destroys the first context instead of leaking it
    assert g._h.value is not None
    g.Compute()
    g.close()
    assert first is not None
