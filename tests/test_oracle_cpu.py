"""CPU tests: the oracle against the golden vectors and the reference's own Tree.cpp."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _grid_problem(g):
    R, C, k, seed = int(g["R"]), int(g["C"]), int(g["k"]), int(g["seed"])
    return R, C, k, seed


def _edge_oracle(R, C, k, seed, pairwise):
    un, W = O.synth_problem(R, C, k, seed, pairwise)
    el = O.grid_edges(R, C)
    o = O.EdgeOracle(R * C, el, k, un, np.tile(W.reshape(-1), el.size // 2))
    o.init()
    return o


@pytest.mark.parametrize("name,pairwise", [("edge_3x3_k2.npz", "potts"), ("edge_8x8_k4.npz", "potts"),
                                           ("edge_6x5_k8_trunc.npz", "trunc_linear")])
def test_edge_oracle_matches_reference_vectors(name, pairwise):
    g = np.load(os.path.join(GOLD, name))
    R, C, k, seed = _grid_problem(g)
    o = _edge_oracle(R, C, k, seed, pairwise)
    assert np.array_equal(o.vertcnt, g["vertcnt"])
    assert np.array_equal(o.state()["pot_node"], g["pot_node"])
    assert np.array_equal(o.state()["pot_edge"], g["pot_edge"])
    objs = g["objective"]
    for it in range(1, objs.size + 1):
        o.compute()
        o.update()
        assert o.objective() == objs[it - 1], "objective differs at iteration %d" % it
        st = o.state()
        for n in ("mu_node", "lambda_node", "mu_edge", "m_node", "vertmu"):
            key = "it%d_%s" % (it, n)
            if key in g:
                assert np.array_equal(st[n], g[key]), key


def test_edge_oracle_objective_trace_64x64():
    g = np.load(os.path.join(GOLD, "edge_64x64_k4_obj.npz"))
    known = json.load(open(os.path.join(GOLD, "known_answers.json")))["64x64_k4"]
    o = _edge_oracle(64, 64, 4, 1234, "potts")
    objs = g["objective"]
    for it in range(1, 101):  # the first 100 of the 500 reference iterations
        o.compute()
        o.update()
        v = o.objective()
        assert v == objs[it - 1]
        if str(it) in known:
            assert "%.6f" % v == "%.6f" % known[str(it)]
    # the stored trace carries the reference's printed values to the end
    for it_s, val in known.items():
        assert "%.6f" % objs[int(it_s) - 1] == "%.6f" % val


def test_known_answers_3x3():
    known = json.load(open(os.path.join(GOLD, "known_answers.json")))["3x3_k2"]
    o = _edge_oracle(3, 3, 2, 1234, "potts")
    assert list(o.vertcnt) == known["vertCnt"]
    assert ["%.9g" % x for x in o.state()["pot_node"][:4]] == ["%.9g" % x for x in known["edge0_x_potential_node"]]
    for it in (1, 2, 3):
        o.compute()
        o.update()
        assert "%.9f" % o.objective() == "%.9f" % known[str(it)]
    st = o.state()
    assert ["%.9g" % x for x in st["vertmu"][8:10]] == ["%.9g" % x for x in known["it3_m_v4"]]
    assert ["%.9g" % x for x in st["mu_node"][:4]] == ["%.9g" % x for x in known["it3_edge0_mu"]]
    assert ["%.9g" % x for x in st["lambda_node"][:4]] == ["%.9g" % x for x in known["it3_edge0_lambda"]]


def test_edge_oracle_on_general_graph():
    g = np.load(os.path.join(GOLD, "edge_graph7_k3.npz"))
    o = O.EdgeOracle(int(g["V"]), g["edge_list"], int(g["k"]), g["unary"], g["edge_pot"])
    o.init()
    assert np.array_equal(o.vertcnt, g["vertcnt"])
    for it in range(g["objective"].size):
        o.compute()
        o.update()
        assert o.objective() == g["objective"][it]
    st = o.state()
    for n in ("mu_node", "lambda_node", "mu_edge", "vertmu"):
        assert np.array_equal(st[n], g["final_" + n])


def test_oracle_bitwise_equals_reference_tree_cpp():
    ref = O.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(5)
    for k in (2, 3, 4, 7, 16):
        for _ in range(20):
            mu = rng.random(2 * k, dtype=np.float32) + 1e-3
            mu[:k] /= mu[:k].sum(); mu[k:] /= mu[k:].sum()
            mue = rng.random(k * k, dtype=np.float32); mue /= mue.sum()
            if rng.random() < 0.3:
                mue[rng.integers(k * k)] = 0.0  # exercises the EPSILON clamp (Tree.cpp:53-56)
            m = rng.random(2 * k, dtype=np.float32); m[:k] /= m[:k].sum(); m[k:] /= m[k:].sum()
            pot = rng.random(2 * k, dtype=np.float32)
            lam = (rng.random(2 * k, dtype=np.float32) - 0.5) * 0.1
            pote = rng.random(k * k, dtype=np.float32)
            a = O.edge_solve_one(k, mu, pot, m, lam, mue, pote)
            b = O.edge_solve_one(k, mu, pot, m, lam, mue, pote, fn=ref.ref_solve_one)
            for x, y in zip(a, b):
                assert np.array_equal(x, y)


def test_oracle_loop_bitwise_equals_reference_loop():
    ref = O.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    R, C, k = 9, 7, 5
    un, W = O.synth_problem(R, C, k, 99, "dense")
    el = O.grid_edges(R, C)
    ep, _ = O.lcg_fill(4242, (el.size // 2) * k * k)
    a = O.EdgeOracle(R * C, el, k, un, ep)
    b = O.RefEdge(ref, R * C, el, k, un, ep)
    a.init(); b.init()
    for _ in range(30):
        a.compute(); a.update(); b.compute(); b.update()
        assert a.objective() == b.objective()
    sa, sb = a.state(), b.state()
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), n


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "chain_*.npz"))))
def test_chain_oracle_matches_its_vectors(name):
    g = np.load(os.path.join(GOLD, name))
    R, C, k, seed = _grid_problem(g)
    un, W = O.synth_problem(R, C, k, seed, str(g["pairwise"]))
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    for it in range(g["primal"].size):
        o.compute()
        o.update()
        p, d, l = o.objective()
        assert p == g["primal"][it] and d == g["dual"][it] and l == g["linear"][it]
        assert d <= p * (1 + 1e-6) + 1e-6, "the dual bound exceeds the primal energy"
    assert np.array_equal(o.xrow, g["xrow"]) and np.array_equal(o.xcol, g["xcol"])
    assert np.array_equal(o.labels(), g["labels"])
    assert np.array_equal(o.lambda_row, g["lambda_row"])
    # the two copies' multipliers cancel exactly
    assert np.array_equal(o.lambda_row, -o.lambda_col)


def test_chain_oracle_against_brute_force():
    """Viterbi with lowest-index ties against exhaustive search on tiny chains."""
    import itertools
    rng = np.random.default_rng(11)
    for k, T in ((2, 5), (3, 4), (4, 3), (3, 1), (3, 2), (2, 9), (3, 6), (2, 10), (5, 3)):
        # quantised costs so that exact ties occur
        un = (rng.integers(0, 4, size=(1, T, k)) * 0.25).astype(np.float32)
        W = (rng.integers(0, 3, size=(k, k)) * 0.25).astype(np.float32)
        o = O.ChainOracle(1, T, k, un, W)
        o.init()
        o.compute(0, 1, 0, 0)  # the single row chain only
        got = o.xrow[0]
        th = (un[0] / np.float32(2)).astype(np.float32)
        best = None
        for lab in itertools.product(range(k), repeat=T):
            val = sum(float(th[t, lab[t]]) for t in range(T)) + sum(float(W[lab[t], lab[t + 1]]) for t in range(T - 1))
            if best is None or val < best[0] - 1e-9:
                best = (val, lab)
        got_val = sum(float(th[t, got[t]]) for t in range(T)) + sum(float(W[got[t], got[t + 1]]) for t in range(T - 1))
        assert abs(got_val - best[0]) < 1e-6


def test_edge_ranks_oracle_one_rank_is_the_plain_oracle():
    R, C, k = 7, 6, 3
    un, W = O.synth_problem(R, C, k, 5, "dense")
    el = O.grid_edges(R, C)
    ep = np.tile(W.reshape(-1), el.size // 2)
    a = O.EdgeOracle(R * C, el, k, un, ep)
    b = O.EdgeRanksOracle(R * C, el, k, un, ep, 1)
    a.init(); b.init()
    for _ in range(10):
        a.compute(); a.update()
        b.compute(); b.update()
    sa, sb = a.state(), b.state()
    for n in sb:
        assert np.array_equal(sa[n], sb[n]), n
    assert a.objective() == b.objective()


@pytest.mark.parametrize("ranks", [2, 4])
def test_edge_ranks_oracle_reproduces_the_reference_on_several_ranks(ranks):
    """totalObj at iterations 100 and 500 as the unmodified reference printed it under
    mpiexec -n 2 and -n 4: pins the partition, the neighbour plan and the order in which
    a vertex's partial sums are added (Grid.cpp:445-454, 198-331, 163-181)."""
    known = json.load(open(os.path.join(GOLD, "known_answers.json")))["64x64_k4_%dranks" % ranks]
    R = C = 64
    k = 4
    un, W = O.synth_problem(R, C, k, 1234, "potts")
    el = O.grid_edges(R, C)
    o = O.EdgeRanksOracle(R * C, el, k, un, np.tile(W.reshape(-1), el.size // 2), ranks)
    o.init()
    for it in range(1, 501):
        o.compute(); o.update()
        if str(it) in known:
            assert "%.6f" % o.objective() == "%.6f" % known[str(it)], (it, o.objective(), known[str(it)])
