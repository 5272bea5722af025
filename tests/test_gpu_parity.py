"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and
the golden vectors.  Integer results (labels) and the binary32 shell (multipliers,
consensus) must be bit-exact; sums accumulated in binary64 in a different order and the
edge solver's double-precision exp/log carry a stated tolerance."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OBJ_RTOL = 1e-9      # binary64 sums in a different order
EDGE_RTOL = 1e-5     # double exp/log of the device library vs glibc, rounded to float


def _chain_grid(R, C, k, un, W, batch=1, rank=0, nprocs=1, exchange=0):
    import admm_amd as A
    g = A.Grid(rank=rank, nprocs=nprocs)
    g.exchange = exchange
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k, batch=batch)
    g.Load(un, W)
    g.InitOptimization()
    return g


def _step(g):
    g.Compute(); g.Communicate(); g.Update()


def _check_chain_against_oracle(g, o, R, C, k):
    assert np.array_equal(g.State(13).reshape(R, C), o.xrow)
    assert np.array_equal(g.State(14).reshape(R, C), o.xcol)
    assert np.array_equal(g.Labels(), o.labels())
    assert np.array_equal(g.Consensus().reshape(R, C, k), o.consensus)
    assert np.array_equal(g.State(10).reshape(R, C, k), o.lambda_row)
    assert np.array_equal(g.State(11).reshape(R, C, k), o.lambda_col)
    p, d = g.Objective()
    lin = g.ComputeLinearObj()
    po, do, lo = o.objective()
    assert abs(p - po) <= OBJ_RTOL * abs(po)
    assert abs(d - do) <= OBJ_RTOL * abs(do)
    assert abs(lin - lo) <= OBJ_RTOL * abs(lo)


@pytest.mark.parametrize("R,C,k,pw", [
    (8, 8, 4, "potts"), (64, 64, 4, "potts"), (33, 70, 8, "potts"), (40, 130, 16, "trunc_linear"),
    (70, 65, 32, "potts"), (66, 129, 32, "dense"), (20, 67, 64, "dense"), (9, 200, 5, "dense"),
    (5, 3, 2, "potts"), (1, 40, 4, "potts"), (40, 1, 4, "potts"), (1, 1, 3, "dense"), (3, 300, 21, "dense"),
    (130, 3, 64, "potts"), (2, 2, 33, "dense")])
def test_chain_every_iteration_bit_exact(R, C, k, pw):
    from oracle import oracle as O
    un, W = O.synth_problem(R, C, k, 1234, pw)
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    g = _chain_grid(R, C, k, un, W)
    assert np.array_equal(g.State(12).reshape(R, C, k), o.theta)
    lin0 = g.ComputeLinearObj()  # before the first Compute: labels undefined -> label 0
    assert abs(lin0 - o.objective()[2]) <= OBJ_RTOL * abs(lin0)
    for it in range(6):
        o.compute(); o.update()
        _step(g)
        assert np.array_equal(g.State(13).reshape(R, C), o.xrow), "row-chain labels, iteration %d" % (it + 1)
        assert np.array_equal(g.State(14).reshape(R, C), o.xcol), "column-chain labels, iteration %d" % (it + 1)
    _check_chain_against_oracle(g, o, R, C, k)
    g.close()


@pytest.mark.parametrize("ns", ["1", "2"])
def test_chain_k32_both_lane_layouts(ns, monkeypatch):
    """k = 32 has two kernels: one chain per wavefront split over both halves, and two chains per wavefront."""
    from oracle import oracle as O
    monkeypatch.setenv("ADMM_SWEEP_NS", ns)
    R, C, k = 37, 150, 32
    un, W = O.synth_problem(R, C, k, 7, "dense")
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    g = _chain_grid(R, C, k, un, W)
    for _ in range(5):
        o.compute(); o.update(); _step(g)
    _check_chain_against_oracle(g, o, R, C, k)
    g.close()


def test_chain_ties_lowest_index():
    """All-equal costs: every candidate ties, so every rule about ties is exercised."""
    from oracle import oracle as O
    for (R, C, k) in ((6, 70, 32), (5, 9, 8), (4, 66, 64)):
        un = np.zeros((R * C, k), dtype=np.float32)
        W = np.zeros((k, k), dtype=np.float32)
        o = O.ChainOracle(R, C, k, un, W); o.init()
        g = _chain_grid(R, C, k, un, W)
        for _ in range(3):
            o.compute(); o.update(); _step(g)
        assert np.array_equal(g.Labels(), o.labels())
        assert g.Labels().max() == 0
        g.close()
        # quantised costs: many exact ties between distinct paths
        rng = np.random.default_rng(R * 100 + k)
        un = (rng.integers(0, 3, size=(R * C, k)) * 0.25).astype(np.float32)
        W = (rng.integers(0, 3, size=(k, k)) * 0.25).astype(np.float32)
        o = O.ChainOracle(R, C, k, un, W); o.init()
        g = _chain_grid(R, C, k, un, W)
        for _ in range(5):
            o.compute(); o.update(); _step(g)
        _check_chain_against_oracle(g, o, R, C, k)
        g.close()


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "chain_*.npz"))))
def test_chain_golden(name):
    from oracle import oracle as O
    gold = np.load(os.path.join(GOLD, name))
    R, C, k, seed = int(gold["R"]), int(gold["C"]), int(gold["k"]), int(gold["seed"])
    un, W = O.synth_problem(R, C, k, seed, str(gold["pairwise"]))
    g = _chain_grid(R, C, k, un, W)
    for it in range(gold["primal"].size):
        _step(g)
        lin = g.ComputeLinearObj()
        assert abs(lin - gold["linear"][it]) <= OBJ_RTOL * abs(lin)
    p, d = g.Objective()
    assert abs(p - gold["primal"][-1]) <= OBJ_RTOL * abs(p) and abs(d - gold["dual"][-1]) <= OBJ_RTOL * abs(d)
    assert np.array_equal(g.Labels(), gold["labels"])
    assert np.array_equal(g.State(13).reshape(R, C), gold["xrow"])
    assert np.array_equal(g.State(14).reshape(R, C), gold["xcol"])
    assert np.array_equal(g.State(10).reshape(R, C, k), gold["lambda_row"])
    g.close()


def test_chain_batch_of_independent_grids():
    from oracle import oracle as O
    import admm_amd as A
    B, R, C, k = 5, 24, 40, 8
    uns, oracles = [], []
    W = O.pairwise_table("potts", k)
    for b in range(B):
        un, _ = O.lcg_fill(1234 + b, R * C * k)
        uns.append(un)
        o = O.ChainOracle(R, C, k, un, W); o.init(); oracles.append(o)
    g = _chain_grid(R, C, k, np.concatenate(uns), W, batch=B)
    # Grid.Generate draws the same potentials
    g2 = A.Grid(); g2.SetPartitioner(A.ROW); g2.SetGlobalInfo(R, C, k, batch=B); g2.Generate(seed=1234, pairwise="potts")
    assert np.array_equal(g2._unary, np.concatenate(uns))
    for _ in range(5):
        for o in oracles:
            o.compute(); o.update()
        _step(g)
    lab = g.Labels()
    lam = g.State(10).reshape(B, R, C, k)
    for b, o in enumerate(oracles):
        assert np.array_equal(lab[b], o.labels())
        assert np.array_equal(lam[b], o.lambda_row)
    p, d = g.Objective()
    lin = g.ComputeLinearObj()
    ref = np.array([o.objective() for o in oracles]).sum(axis=0)
    assert abs(p - ref[0]) <= OBJ_RTOL * abs(p) and abs(d - ref[1]) <= OBJ_RTOL * abs(d) and abs(lin - ref[2]) <= OBJ_RTOL * abs(lin)
    g.close()


@pytest.mark.parametrize("exchange", [0, 1], ids=["labels", "dense"])
@pytest.mark.parametrize("G,R,C,k,pw", [(2, 20, 24, 8, "potts"), (3, 17, 10, 32, "dense"), (4, 9, 70, 4, "potts"),
                                        (8, 64, 64, 16, "trunc_linear"), (5, 3, 7, 5, "dense"), (2, 6, 5, 2, "potts")])
def test_chain_strips_equal_single_strip(G, R, C, k, pw, exchange):
    """The grid cut into G row / column strips (all in this process, on one GPU) gives the
    single-strip result bit for bit; strips may be empty when G exceeds rows or columns.
    Both wire formats: labels, and the float k-vectors of the marginals."""
    from oracle import oracle as O
    import admm_amd as A
    un, W = O.synth_problem(R, C, k, 1234, pw)
    o = O.ChainOracle(R, C, k, un, W); o.init()
    grids = [_chain_grid(R, C, k, un, W, rank=r, nprocs=G, exchange=exchange) for r in range(G)]
    A.communicate_group(grids)  # before the first solve: nothing to tell, must change nothing
    for it in range(6):
        o.compute(); o.update()
        for g in grids:
            g.Compute()
        A.communicate_group(grids)
        for g in grids:
            g.Update()
    lab = np.concatenate([g.Labels().reshape(-1, C) for g in grids], axis=0)
    assert np.array_equal(lab, o.labels())
    prim = dual = lin = 0.0
    for g in grids:
        r0, nr, c0, nc = g.StripInfo()
        assert np.array_equal(g.State(10).reshape(nr, C, k), o.lambda_row[r0:r0 + nr])
        assert np.array_equal(g.State(11).reshape(R, nc, k), o.lambda_col[:, c0:c0 + nc])
        assert np.array_equal(g.State(14).reshape(R, nc), o.xcol[:, c0:c0 + nc])
        assert np.array_equal(g.Consensus().reshape(nr, C, k), o.consensus[r0:r0 + nr])
        p, d = g.Objective()
        prim += p; dual += d; lin += g.ComputeLinearObj()  # summed over strips as admm.cpp:95 sums over ranks
    po, do, lo = o.objective()
    assert abs(prim - po) <= OBJ_RTOL * abs(po) and abs(dual - do) <= OBJ_RTOL * abs(do) and abs(lin - lo) <= OBJ_RTOL * abs(lo)
    for g in grids:
        g.close()


def test_chain_queued_objectives_match_blocking_calls():
    """A queued objective gets its unary term from the next sweep, or from a gather when it
    is read (or queued again) before that; either way it equals the blocking call."""
    from oracle import oracle as O
    R, C, k = 30, 45, 16
    un, W = O.synth_problem(R, C, k, 5, "trunc_linear")
    o = O.ChainOracle(R, C, k, un, W); o.init()
    g = _chain_grid(R, C, k, un, W)
    blocking, ref = [], []
    for it in range(7):
        _step(g)
        o.compute(); o.update()
        ref.append(o.objective()[2])
        g.ObjectiveEnqueue(1 + it)
        if it == 2:
            g.ObjectiveEnqueue(20)   # the same solve queued twice
        if it == 4:                  # read while the unary term is still outstanding
            assert abs(g.ObjectiveRead(5, 1)[0] - ref[-1]) <= OBJ_RTOL * abs(ref[-1])
        if it % 2 == 0:
            blocking.append(g.ComputeLinearObj())
            assert abs(blocking[-1] - ref[-1]) <= OBJ_RTOL * abs(ref[-1])
    q = g.ObjectiveRead(1, 7)        # the last one is completed by the read
    assert np.all(np.abs(q - np.array(ref)) <= OBJ_RTOL * np.abs(ref))
    assert abs(g.ObjectiveRead(20, 1)[0] - ref[2]) <= OBJ_RTOL * abs(ref[2])
    g.close()


@pytest.mark.parametrize("R,C,k,pw,B", [(70, 65, 32, "potts", 1), (20, 67, 64, "dense", 1), (33, 70, 8, "potts", 3),
                                        (9, 200, 5, "dense", 1), (40, 130, 16, "trunc_linear", 1)])
def test_chain_queued_objective_every_lane_layout(R, C, k, pw, B):
    """The sweep's own unary sums (every lane layout, ragged sizes, a batch) against the gather."""
    from oracle import oracle as O
    W = O.pairwise_table(pw, k, 99) if pw == "dense" else O.pairwise_table(pw, k)
    un = np.concatenate([O.lcg_fill(77 + b, R * C * k)[0] for b in range(B)])
    g = _chain_grid(R, C, k, un, W, batch=B)
    blocking = []
    for it in range(4):
        _step(g)
        g.ObjectiveEnqueue(1 + it)
        blocking.append(g.ComputeLinearObj())
    _step(g)                         # supplies the fourth unary term
    q = g.ObjectiveRead(1, 4)
    assert np.all(np.abs(q - np.array(blocking)) <= OBJ_RTOL * np.abs(blocking))
    g.close()


# ---------------------------------------------------------------- edge solver (the reference's algorithm)

def _edge_grid(R, C, k, un, W, per_edge=False):
    import admm_amd as A
    g = A.Grid()
    g.SetPartitioner(A.EDGE)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k)
    g.Load(un, W, per_edge=per_edge)
    g.InitOptimization()
    return g


@pytest.mark.parametrize("name,pw", [("edge_3x3_k2.npz", "potts"), ("edge_8x8_k4.npz", "potts"),
                                      ("edge_6x5_k8_trunc.npz", "trunc_linear")])
def test_edge_against_reference_vectors(name, pw):
    from oracle import oracle as O
    gold = np.load(os.path.join(GOLD, name))
    R, C, k, seed = int(gold["R"]), int(gold["C"]), int(gold["k"]), int(gold["seed"])
    un, W = O.synth_problem(R, C, k, seed, pw)
    g = _edge_grid(R, C, k, un, W)
    # the split of the potentials is pure binary32 arithmetic: bit-exact
    assert np.array_equal(g.State(1), gold["pot_node"])
    assert np.array_equal(g.State(5), gold["pot_edge"])
    for it in range(1, gold["objective"].size + 1):
        _step(g)
        obj = g.ComputeLinearObj()
        assert abs(obj - gold["objective"][it - 1]) <= EDGE_RTOL * abs(obj), it
        for which, n in ((0, "mu_node"), (3, "lambda_node"), (4, "mu_edge"), (2, "m_node")):
            key = "it%d_%s" % (it, n)
            if key in gold:
                np.testing.assert_allclose(g.State(which), gold[key], rtol=EDGE_RTOL, atol=1e-7, err_msg=key)
        if "it%d_vertmu" % it in gold:
            np.testing.assert_allclose(g.Consensus().reshape(-1), gold["it%d_vertmu" % it], rtol=EDGE_RTOL, atol=1e-7)
    g.close()


def test_edge_objective_trace_64x64_500_iterations():
    """The reference's own plumbing case: 64 x 64, k = 4, 500 iterations (admm.cpp:74)."""
    from oracle import oracle as O
    gold = np.load(os.path.join(GOLD, "edge_64x64_k4_obj.npz"))
    un, W = O.synth_problem(64, 64, 4, 1234, "potts")
    g = _edge_grid(64, 64, 4, un, W)
    worst = 0.0
    for it in range(500):
        _step(g)
        obj = g.ComputeLinearObj()
        worst = max(worst, abs(obj - gold["objective"][it]) / abs(obj))
    assert worst <= EDGE_RTOL, worst
    assert "%.3f" % obj == "2015.785"
    g.close()


@pytest.mark.parametrize("R,C,k", [(5, 4, 2), (6, 6, 3), (7, 5, 7), (4, 9, 16), (5, 5, 32), (3, 4, 64), (1, 6, 4), (6, 1, 4)])
def test_edge_against_oracle_per_edge_tables(R, C, k):
    from oracle import oracle as O
    un, _ = O.synth_problem(R, C, k, 31, "potts")
    el = O.grid_edges(R, C)
    E = el.size // 2
    ep, _ = O.lcg_fill(777, E * k * k)
    o = O.EdgeOracle(R * C, el, k, un, ep); o.init()
    g = _edge_grid(R, C, k, un, ep, per_edge=True)
    for it in range(12):
        o.compute(); o.update(); _step(g)
        a, b = o.objective(), g.ComputeLinearObj()
        assert abs(a - b) <= EDGE_RTOL * abs(a)
    st = o.state()
    for which, n in ((0, "mu_node"), (3, "lambda_node"), (4, "mu_edge"), (2, "m_node")):
        np.testing.assert_allclose(g.State(which), st[n], rtol=EDGE_RTOL, atol=1e-7, err_msg=n)
    g.close()


@pytest.mark.parametrize("penalty,prime,rho", [(0.5, 1.0, 1.0), (0.7, 0.3, 1.0), (1.5, 2.0, 0.5), (0.25, 0.75, 2.0), (0.1, 0.5, 1.0)])
@pytest.mark.parametrize("k", [4, 16, 32])
def test_edge_penalties(penalty, prime, rho, k):
    """Other penalties than the drivers' (Grid.cpp:79-80 are the class defaults, the last case):
    x / penaltyPrime is a multiplication for a power of two and a product with the rounded
    reciprocal otherwise (Tree.cpp:160, 172)."""
    import admm_amd as A
    from oracle import oracle as O
    R, C = 6, 7
    un, _ = O.synth_problem(R, C, k, 77, "potts")
    el = O.grid_edges(R, C)
    ep, _ = O.lcg_fill(4242, (el.size // 2) * k * k)
    o = O.EdgeOracle(R * C, el, k, un, ep, penalty=penalty, penalty_prime=prime, rho=rho); o.init()
    g = A.Grid()
    g.SetPartitioner(A.EDGE); g.SetPenalties(penalty, prime); g.SetRhoEdge(rho)
    g.SetGlobalInfo(R, C, k); g.Load(un, ep, per_edge=True); g.InitOptimization()
    for it in range(10):
        o.compute(); o.update(); _step(g)
        a, b = o.objective(), g.ComputeLinearObj()
        assert abs(a - b) <= EDGE_RTOL * abs(a), it
    st = o.state()
    for which, n in ((0, "mu_node"), (3, "lambda_node"), (4, "mu_edge"), (2, "m_node")):
        np.testing.assert_allclose(g.State(which), st[n], rtol=EDGE_RTOL, atol=1e-7, err_msg=n)
    g.close()


def test_edge_on_general_graph():
    import admm_amd as A
    gold = np.load(os.path.join(GOLD, "edge_graph7_k3.npz"))
    g = A.Grid()
    g.SetPartitioner(A.EDGE)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(1, 1, int(gold["k"]))
    g.SetGraph(int(gold["V"]), gold["edge_list"])
    g.Load(gold["unary"], gold["edge_pot"], per_edge=True)
    g.InitOptimization()
    for it in range(gold["objective"].size):
        _step(g)
        obj = g.ComputeLinearObj()
        assert abs(obj - gold["objective"][it]) <= EDGE_RTOL * abs(obj)
    np.testing.assert_allclose(g.State(0), gold["final_mu_node"], rtol=EDGE_RTOL, atol=1e-7)
    np.testing.assert_allclose(g.Consensus().reshape(-1), gold["final_vertmu"], rtol=EDGE_RTOL, atol=1e-7)
    g.close()


def test_edge_clamps_non_positive_marginals():
    """Zero edge potentials make marginals underflow to 0; the next Compute must clamp them (Tree.cpp:46-57)."""
    from oracle import oracle as O
    R, C, k = 3, 3, 4
    un = np.zeros((R * C, k), dtype=np.float32)
    un[:, 0] = 200.0  # exp(-200) underflows in binary32
    W = O.pairwise_table("potts", k)
    el = O.grid_edges(R, C)
    o = O.EdgeOracle(R * C, el, k, un, np.tile(W.reshape(-1), el.size // 2)); o.init()
    g = _edge_grid(R, C, k, un, W)
    for _ in range(4):
        o.compute(); o.update(); _step(g)
    assert (o.state()["mu_node"] == 0).any(), "the case does not reach the clamp"
    np.testing.assert_allclose(g.State(0), o.state()["mu_node"], rtol=EDGE_RTOL, atol=1e-30)
    a, b = o.objective(), g.ComputeLinearObj()
    assert abs(a - b) <= EDGE_RTOL * abs(a)
    g.close()


@pytest.mark.parametrize("R,C,k", [(3, 3, 2), (9, 7, 3), (17, 5, 8), (6, 11, 16), (13, 9, 32), (5, 7, 64), (40, 33, 5)])
def test_edge_objective_summed_by_the_solve_equals_the_separate_pass(R, C, k, monkeypatch):
    """The solve sums the linear objective of the marginals it writes (one term per workgroup);
    ADMM_EDGE_FUSED_OBJ=0 keeps the separate pass over marginals and potentials.  The same
    binary32 products in another binary64 order; before the first solve both are the separate pass."""
    from oracle import oracle as O
    un, _ = O.synth_problem(R, C, k, 5, "potts")
    E = O.grid_edges(R, C).size // 2
    ep, _ = O.lcg_fill(99, E * k * k)
    monkeypatch.setenv("ADMM_EDGE_FUSED_OBJ", "0")
    gs = _edge_grid(R, C, k, un, ep, per_edge=True)
    monkeypatch.delenv("ADMM_EDGE_FUSED_OBJ")
    gf = _edge_grid(R, C, k, un, ep, per_edge=True)
    assert gs.ComputeLinearObj() == gf.ComputeLinearObj()
    for it in range(6):
        _step(gs); _step(gf)
        a, b = gs.ComputeLinearObj(), gf.ComputeLinearObj()
        assert abs(a - b) <= 1e-12 * abs(a), (it, a, b)
        assert b == gf.ComputeLinearObj()        # asked twice: the same sum
    for which in (0, 3, 4):
        assert np.array_equal(gs.State(which), gf.State(which))
    gs.close(); gf.close()


@pytest.mark.parametrize("R,C,k,scale", [(9, 7, 4, 1.0), (6, 8, 16, 4.0), (7, 5, 32, 30.0), (4, 4, 64, 1.0), (8, 9, 11, 120.0)])
def test_edge_state_matches_the_oracle_to_the_last_bit(R, C, k, scale):
    """The tolerance of the edge solver (1e-5) is what its transcendental functions guarantee
    against glibc's; what they deliver is the oracle's binary32 state bit for bit (every value
    rounded to binary32 right after a binary64 exp or log that is within an ulp of glibc's).  A
    handful of values one binary32 step apart would still be within that guarantee: none, or
    nearly none, is what this test pins."""
    from oracle import oracle as O
    rng = np.random.default_rng(R * 100 + k)
    un = (rng.random((R * C, k), dtype=np.float32) * scale).astype(np.float32)
    el = O.grid_edges(R, C)
    E = el.size // 2
    ep = (rng.random(E * k * k, dtype=np.float32) * min(scale, 30.0)).astype(np.float32)
    o = O.EdgeOracle(R * C, el, k, un, ep); o.init()
    g = _edge_grid(R, C, k, un, ep, per_edge=True)
    for _ in range(8):
        o.compute(); o.update(); _step(g)
    st = o.state()
    differing = 0
    for which, n in ((0, "mu_node"), (3, "lambda_node"), (4, "mu_edge"), (2, "m_node")):
        a, b = g.State(which), st[n].reshape(-1)
        ne = a.view(np.uint32) != b.view(np.uint32)
        differing += int(ne.sum())
        if ne.any():  # one binary32 step at most
            assert np.all(np.abs(a[ne].view(np.int32).astype(np.int64) - b[ne].view(np.int32).astype(np.int64)) <= 1), n
    assert differing <= 3, differing
    g.close()


def test_edge_objective_after_a_second_init():
    """admm_init after solves starts over: the objective asked for before the next solve is the
    initial state's, not the sum the last solve left behind."""
    from oracle import oracle as O
    R, C, k = 9, 8, 8
    un, W = O.synth_problem(R, C, k, 3, "potts")
    g = _edge_grid(R, C, k, un, W)
    first = g.ComputeLinearObj()
    for _ in range(3):
        _step(g)
    later = g.ComputeLinearObj()
    from admm_amd._lib import lib
    g._ck(lib.admm_init(g._h))
    again = g.ComputeLinearObj()
    assert again == first and again != later
    _step(g)
    o = O.EdgeOracle(R * C, O.grid_edges(R, C), k, un, np.tile(W.reshape(-1), O.grid_edges(R, C).size // 2)); o.init()
    o.compute(); o.update()
    assert abs(g.ComputeLinearObj() - o.objective()) <= EDGE_RTOL * abs(o.objective())
    g.close()


# ---------------------------------------------------------------- call-order errors

def test_call_order_errors():
    import ctypes as C
    import admm_amd as A
    from admm_amd._lib import AdmmDesc, lib
    d = AdmmDesc()
    d.rows, d.cols, d.k, d.batch, d.solver = 4, 4, 4, 1, 1
    d.penalty, d.penalty_prime, d.rho, d.nranks = 0.5, 1.0, 1.0, 1
    h = C.c_void_p()
    assert lib.admm_create(C.byref(d), C.byref(h)) == 0
    assert lib.admm_compute(h) == -2 and b"admm_init" in lib.admm_last_error(h)
    assert lib.admm_init(h) == -2
    un = np.zeros(64, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    assert lib.admm_upload_unaries(h, un.ctypes.data_as(fp)) == 0
    assert lib.admm_upload_pairwise(h, un.ctypes.data_as(fp), 1) == -1  # the chain solver shares one table
    assert lib.admm_upload_pairwise(h, un.ctypes.data_as(fp), 0) == 0
    assert lib.admm_init(h) == 0
    assert lib.admm_objective_enqueue(h, 1) == -2  # no Compute yet
    assert lib.admm_compute(h) == 0 and lib.admm_communicate(h) == 0 and lib.admm_update(h) == 0
    assert lib.admm_objective_enqueue(h, 0) == -1
    assert lib.admm_destroy(h) == 0
    d.nranks, d.rank = 2, 1
    assert lib.admm_create(C.byref(d), C.byref(h)) == 0
    lib.admm_upload_unaries(h, un.ctypes.data_as(fp)); lib.admm_upload_pairwise(h, un.ctypes.data_as(fp), 0); lib.admm_init(h)
    assert lib.admm_compute(h) == 0
    assert lib.admm_communicate(h) == -4 and b"admm_comm_init" in lib.admm_last_error(h)
    lib.admm_destroy(h)
