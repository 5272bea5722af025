"""The edge solver beyond 64 labels (the reference's class Edge allocates for any k,
Tree.cpp:15-22): one edge per wavefront, the k x k table in HBM.  Checked against the
oracle like the kernels for up to 64 labels: binary32 state bit for bit (a handful of
values one binary32 step apart would be within what the transcendental functions
guarantee), objective to 1e-12."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _grid(R, C, k, un, ep, penalties=(0.5, 1.0), rho=1.0, edges=None, V=None):
    import admm_amd as A
    g = A.Grid()
    g.SetPartitioner(A.EDGE)
    g.SetPenalties(*penalties)
    g.SetRhoEdge(rho)
    g.SetGlobalInfo(R, C, k)
    if edges is not None:
        g.SetGraph(V, edges)
    g.Load(un, ep, per_edge=True)
    g.InitOptimization()
    return g


def _compare(g, o, iters):
    for _ in range(iters):
        o.compute(); o.update()
        g.Compute(); g.Communicate(); g.Update()
    st = o.state()
    differing = 0
    for which, n in ((0, "mu_node"), (3, "lambda_node"), (4, "mu_edge"), (2, "m_node")):
        a, b = g.State(which), st[n].reshape(-1)
        ne = a.view(np.uint32) != b.view(np.uint32)
        differing += int(ne.sum())
        if ne.any():
            assert np.all(np.abs(a[ne].view(np.int32).astype(np.int64) - b[ne].view(np.int32).astype(np.int64)) <= 1), n
    assert differing <= 3, differing
    a, b = o.objective(), g.ComputeLinearObj()
    assert abs(a - b) <= 1e-12 * abs(a), (a, b)


@pytest.mark.parametrize("R,C,k,scale", [(3, 3, 65, 1.0), (4, 3, 100, 2.0), (3, 2, 128, 1.0), (2, 3, 200, 8.0),
                                         (2, 2, 300, 1.0), (1, 2, 1024, 1.0)])
def test_edge_solver_with_more_labels_than_lanes(R, C, k, scale):
    from oracle import oracle as O
    rng = np.random.default_rng(k)
    un = (rng.random((R * C, k), dtype=np.float32) * scale).astype(np.float32)
    el = O.grid_edges(R, C)
    E = el.size // 2
    ep = (rng.random(E * k * k, dtype=np.float32) * scale).astype(np.float32)
    o = O.EdgeOracle(R * C, el, k, un, ep); o.init()
    g = _grid(R, C, k, un, ep)
    assert g.LaunchInfo() == {}
    _compare(g, o, 4)
    assert g.LaunchInfo()["objective_summed_by_the_solve"] == 1
    g.close()


def test_many_labels_general_penalties_and_a_graph():
    """penaltyPrime and rho that are neither 1 nor powers of two (the divisions of
    Tree.cpp:160, 172 as divisions), on a graph that is not a grid, with enough edges
    for several workgroups and a last one that is not full."""
    from oracle import oracle as O
    V, k = 9, 70
    el = np.array([0, 1, 1, 2, 2, 0, 2, 3, 3, 4, 4, 5, 5, 3, 5, 6, 6, 7, 7, 8, 8, 6, 0, 8, 1, 7], dtype=np.int32)
    E = el.size // 2
    rng = np.random.default_rng(7)
    un = rng.random((V, k), dtype=np.float32)
    ep = rng.random(E * k * k, dtype=np.float32)
    o = O.EdgeOracle(V, el, k, un, ep, penalty=0.3, penalty_prime=0.7, rho=1.3); o.init()
    g = _grid(0, 0, k, un, ep, penalties=(0.3, 0.7), rho=1.3, edges=el, V=V)
    _compare(g, o, 3)
    g.close()
