"""First GPU parity checks: HIP path through the C ABI against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _chain_case(R, C, k, pairwise, iters, seed=1234):
    from oracle import oracle as O
    import admm_amd as A
    un, W = O.synth_problem(R, C, k, seed, pairwise)
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
        xr = g.State(13).reshape(R, C)
        xc = g.State(14).reshape(R, C)
        assert np.array_equal(xr, o.xrow), "row-chain labels differ at iteration %d" % (it + 1)
        assert np.array_equal(xc, o.xcol), "column-chain labels differ at iteration %d" % (it + 1)
    assert np.array_equal(g.Labels(), o.labels())
    assert np.array_equal(g.Consensus().reshape(R, C, k), o.consensus)
    assert np.array_equal(g.State(10).reshape(R, C, k), o.lambda_row)
    assert np.array_equal(g.State(11).reshape(R, C, k), o.lambda_col)
    p, d = g.Objective()
    lin = g.ComputeLinearObj()
    po, do, lo = o.objective()
    assert abs(p - po) <= 1e-9 * abs(po)
    assert abs(d - do) <= 1e-9 * abs(do)
    assert abs(lin - lo) <= 1e-9 * abs(lo)
    g.close()


@pytest.mark.parametrize("R,C,k,pw", [(8, 8, 4, "potts"), (64, 64, 4, "potts"), (33, 70, 8, "potts"),
                                      (40, 130, 16, "trunc_linear"), (70, 65, 32, "potts"),
                                      (20, 67, 64, "dense"), (9, 200, 5, "dense"), (5, 3, 2, "potts")])
def test_chain_parity(R, C, k, pw):
    _chain_case(R, C, k, pw, 6)


def test_edge_parity_64x64_k4():
    from oracle import oracle as O
    import admm_amd as A
    R = C = 64; k = 4
    un, W = O.synth_problem(R, C, k, 1234, "potts")
    el = O.grid_edges(R, C)
    E = el.size // 2
    o = O.EdgeOracle(R * C, el, k, un, np.tile(W.reshape(-1), E))
    o.init()
    g = A.Grid()
    g.SetPartitioner(A.EDGE)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k)
    g.Load(un, W)
    g.InitOptimization()
    assert np.array_equal(g.State(1), o.state()["pot_node"])
    for it in range(20):
        o.compute(); o.update()
        g.Compute(); g.Communicate(); g.Update()
        oo, og = o.objective(), g.ComputeLinearObj()
        assert abs(oo - og) <= 1e-5 * abs(oo), (it, oo, og)
    s = o.state()
    for which, name in [(0, "mu_node"), (3, "lambda_node"), (4, "mu_edge"), (2, "m_node")]:
        np.testing.assert_allclose(g.State(which), s[name], rtol=1e-5, atol=1e-7, err_msg=name)
    g.close()
