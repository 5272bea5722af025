"""The benchmark configurations at full size: the HIP path against the oracle (run with one
chain per host thread) for a few iterations, plus properties that do not need the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _energy_numpy(labels, unary, W):
    """Energy of a labelling, computed from the inputs only (binary64)."""
    R, C = labels.shape
    u = unary.reshape(R, C, -1)
    e = np.take_along_axis(u, labels[..., None].astype(np.int64), axis=2).astype(np.float64).sum()
    e += W[labels[:, :-1], labels[:, 1:]].astype(np.float64).sum()
    e += W[labels[:-1, :], labels[1:, :]].astype(np.float64).sum()
    return e


def _run(R, C, k, pw, iters, seed=1234):
    from oracle import oracle as O
    import admm_amd as A
    O.set_threads(min(16, os.cpu_count() or 1))
    un, W = O.synth_problem(R, C, k, seed, pw)
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k)
    g.Load(un, W)
    g.InitOptimization()
    for it in range(iters):
        o.compute(); o.update()
        g.Compute(); g.Communicate(); g.Update()
    lab = g.Labels()
    assert np.array_equal(lab, o.labels()), "decoded labels differ"
    assert np.array_equal(g.State(13).reshape(R, C), o.xrow)
    assert np.array_equal(g.State(14).reshape(R, C), o.xcol)
    lr = g.State(10).reshape(R, C, k)
    assert np.array_equal(lr, o.lambda_row)
    lc = g.State(11).reshape(R, C, k)
    assert np.array_equal(lc, o.lambda_col)
    # properties
    assert np.array_equal(lr, -lc), "the two copies' multipliers must cancel exactly"
    assert lab.min() >= 0 and lab.max() < k
    p, d = g.Objective()
    lin = g.ComputeLinearObj()
    po, do, lo = o.objective()
    assert abs(p - po) <= 1e-9 * abs(po) and abs(d - do) <= 1e-9 * abs(do) and abs(lin - lo) <= 1e-9 * abs(lo)
    assert d <= p, "dual bound above the primal energy"
    e = _energy_numpy(lab, un, W)
    assert abs(e - p) <= 1e-6 * abs(e), (e, p)  # theta/2 + theta/2 vs theta: exact halves, binary64 sums
    g.close()


def test_config_2048x2048_k32_potts():
    _run(2048, 2048, 32, "potts", 3)


def test_config_1024x1024_k64_dense():
    _run(1024, 1024, 64, "dense", 3)


def test_config_512x512_k16_trunc_linear():
    _run(512, 512, 16, "trunc_linear", 10)


def test_tall_grid_far_offsets():
    """Beyond the benchmark sizes: 6001 x 1500, k = 32 (1.15 GB per operand array, odd chain
    length): the column chains step 192 kB at a time through ~0.6 GB of descriptor range."""
    _run(6001, 1500, 32, "dense", 2, seed=77)


def test_config_batch_256_grids_256x256_k8():
    from oracle import oracle as O
    import admm_amd as A
    O.set_threads(min(16, os.cpu_count() or 1))
    B, R, C, k = 256, 256, 256, 8
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k, batch=B)
    g.Generate(seed=1234, pairwise="potts")
    un_all = g._unary.reshape(B, R * C, k)
    g.InitOptimization()
    iters = 3
    for _ in range(iters):
        g.Compute(); g.Communicate(); g.Update()
    lab = g.Labels()
    lam = g.State(10).reshape(B, R, C, k)
    W = O.pairwise_table("potts", k)
    prim = dual = lin = 0.0
    for b in range(B):
        o = O.ChainOracle(R, C, k, un_all[b], W)
        o.init()
        for _ in range(iters):
            o.compute(); o.update()
        assert np.array_equal(lab[b], o.labels()), "grid %d" % b
        assert np.array_equal(lam[b], o.lambda_row), "grid %d" % b
        po, do, lo = o.objective()
        prim += po; dual += do; lin += lo
    p, d = g.Objective()
    l = g.ComputeLinearObj()
    assert abs(p - prim) <= 1e-9 * abs(prim) and abs(d - dual) <= 1e-9 * abs(dual) and abs(l - lin) <= 1e-9 * abs(lin)
    g.close()
