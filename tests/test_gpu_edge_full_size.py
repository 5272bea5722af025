"""Edge solver at the size of the bench workload (512 x 512, k = 32, Potts): 523 264 edges, whose
restatement on the host would take a minute per iteration on one core.  Checked instead on a
sample of edges against the oracle's solve of one edge (the state before and after admm_compute
is downloaded, Tree.cpp:64-274) and, over all edges, through what every solve must satisfy:
marginals that sum to one, edge marginals whose rows sum to the first endpoint's marginal
(Tree.cpp:256-273), the consensus as the mean of a vertex's copies (Grid.cpp:163-181), and the
linear objective as the sum of the binary32 products (Grid.cpp:1665-1680)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGE_RTOL = 1e-5


def test_full_size_edge_solver_sampled_edges_and_invariants():
    import admm_amd as A
    from oracle import oracle as O
    R = C = 512
    k = 32
    un, W = O.synth_problem(R, C, k, 1234, "potts")
    el = O.grid_edges(R, C).reshape(-1, 2)
    E = el.shape[0]
    g = A.Grid()
    g.SetPartitioner(A.EDGE); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k); g.Load(un, W); g.InitOptimization()
    rng = np.random.default_rng(5)
    sample = np.unique(np.concatenate([[0, 1, E - 1, E // 2], rng.integers(0, E, 120)]))
    pote = g.State(5).reshape(E, k * k)
    pot = g.State(1).reshape(E, 2 * k)
    cnt = np.bincount(el.reshape(-1), minlength=R * C).astype(np.float64)
    for it in range(3):
        mu0 = g.State(0).reshape(E, 2 * k)[sample].copy()
        m0 = g.State(2).reshape(E, 2 * k)[sample].copy()
        lam0 = g.State(3).reshape(E, 2 * k)[sample].copy()
        mue0 = g.State(4).reshape(E, k * k)[sample].copy()
        g.Compute()
        mu1 = g.State(0).reshape(E, 2 * k)
        lam1 = g.State(3).reshape(E, 2 * k)
        mue1 = g.State(4).reshape(E, k * k)
        for n, e in enumerate(sample):
            rmu, rlam, rmue, _, _ = O.edge_solve_one(k, mu0[n], pot[e], m0[n], lam0[n], mue0[n], pote[e])
            np.testing.assert_allclose(mu1[e], rmu, rtol=EDGE_RTOL, atol=1e-9, err_msg="mu, edge %d, iteration %d" % (e, it + 1))
            np.testing.assert_allclose(lam1[e], rlam, rtol=EDGE_RTOL, atol=1e-7, err_msg="lambda, edge %d" % e)
            np.testing.assert_allclose(mue1[e], rmue, rtol=EDGE_RTOL, atol=1e-12, err_msg="edge marginal, edge %d" % e)
        # every edge
        assert np.all(np.isfinite(mu1)) and np.all(np.isfinite(mue1)) and mu1.min() >= 0 and mue1.min() >= 0
        s = mu1.reshape(E, 2, k).sum(axis=2, dtype=np.float64)
        assert np.abs(s - 1.0).max() <= 1e-5
        rows = mue1.reshape(E, k, k).sum(axis=2, dtype=np.float64)
        assert np.abs(rows - mu1[:, :k]).max() <= 2e-6
        want = (pot * mu1).sum(dtype=np.float64) + (pote * mue1).sum(dtype=np.float64)   # binary32 products, binary64 sum
        got = g.ComputeLinearObj()
        assert abs(got - want) <= 1e-9 * abs(want), (got, want)
        g.Communicate(); g.Update()
        mean = np.zeros((R * C, k))
        np.add.at(mean, el.reshape(-1), mu1.reshape(2 * E, k).astype(np.float64))
        mean /= cnt[:, None]
        assert np.abs(g.Consensus().reshape(R * C, k) - mean).max() <= 1e-6
        del mu1, lam1, mue1
    g.close()
