"""Graph file -> solver: the reference's own way in (admm.cpp:66-70:
SetGlobalInfo(fname), Load(fname)), through the Python host and the C++ driver."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADMM = os.path.join(ROOT, "admm_amd", "admm")


def test_chain_solver_from_file_bit_exact(tmp_path):
    import admm_amd as A
    from oracle import oracle as O
    R, C, k = 23, 70, 16
    un, W = O.synth_problem(R, C, k, 5, "trunc_linear")
    src = A.Grid()
    src.SetPartitioner(A.ROW)
    src.SetGlobalInfo(R, C, k)
    src.Load(un, W)
    f = tmp_path / "grid.nc"
    src.Save(f)

    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(f)
    assert (g.rows, g.cols, g.k) == (R, C, k)
    g.Load(f)
    g.InitOptimization()
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    for _ in range(4):
        o.compute(); o.update()
        g.Compute(); g.Communicate(); g.Update()
    assert np.array_equal(g.Labels(), o.labels())
    assert np.array_equal(g.State(10).reshape(R, C, k), o.lambda_row)
    g.close()


def test_edge_solver_on_a_general_graph_from_file(tmp_path):
    """The reference accepts any edge list (Grid.cpp:476-494); per-edge tables."""
    import admm_amd as A
    from admm_amd import graphfile as GF
    from oracle import oracle as O
    rng = np.random.default_rng(9)
    V, k = 40, 5
    el = np.array([(i, (i + 1) % V) for i in range(V)] + [(int(a), int(b)) for a, b in rng.integers(0, V, size=(30, 2)) if a != b],
                  dtype=np.int32)
    E = el.shape[0]
    un = rng.random((V, k), dtype=np.float32)
    pw = rng.random((E, k * k), dtype=np.float32)
    f = tmp_path / "graph.nc"
    GF.write(f, V, el, -un, -pw)  # a file holds scores

    g = A.Grid()
    g.SetPartitioner(A.EDGE)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(f)
    g.Load(f)
    g.InitOptimization()
    o = O.EdgeOracle(V, el.reshape(-1), k, un, pw.reshape(-1))
    o.init()
    for _ in range(8):
        o.compute(); o.update()
        g.Compute(); g.Communicate(); g.Update()
        a, b = o.objective(), g.ComputeLinearObj()
        assert abs(a - b) <= 1e-5 * abs(a)
    np.testing.assert_allclose(g.Consensus(), o.state()["vertmu"].reshape(V, k), rtol=1e-5, atol=1e-7)
    g.close()


def test_chain_solver_rejects_files_it_cannot_take(tmp_path):
    import admm_amd as A
    from admm_amd import graphfile as GF
    R, C, k = 4, 5, 3
    el = A.grid_edges(R, C).reshape(-1, 2)
    rng = np.random.default_rng(0)
    un = rng.random((R * C, k), dtype=np.float32)
    f = tmp_path / "per_edge.nc"
    GF.write(f, R * C, el, un, rng.random((el.shape[0], k * k), dtype=np.float32), rows=R, cols=C)
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetGlobalInfo(f)
    with pytest.raises(ValueError, match="shared table"):
        g.Load(f)
    f2 = tmp_path / "shuffled.nc"
    GF.write(f2, R * C, el[::-1].copy(), un, np.zeros((k, k), dtype=np.float32), rows=R, cols=C)
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetGlobalInfo(f2)
    with pytest.raises(ValueError, match="edge list"):
        g.Load(f2)


def test_cpp_driver_reads_the_file_it_saved(tmp_path):
    """admm <rows> <cols> <k> --save FILE, then admm FILE: same objective trace."""
    f = tmp_path / "saved.nc"
    pick = lambda s: re.findall(r"(totalObj|primal|dual):\s+([-0-9.]+)", s)
    for solver, extra in (("edge", []), ("chain", ["--pairwise", "trunc_linear"])):
        a = subprocess.run([ADMM, "12", "9", "4", "--solver", solver, "--iters", "10", "--save", str(f)] + extra,
                           capture_output=True, text=True, timeout=300)
        assert a.returncode == 0, a.stderr[-2000:]
        b = subprocess.run([ADMM, str(f), "--solver", solver, "--iters", "10"], capture_output=True, text=True, timeout=300)
        assert b.returncode == 0, b.stderr[-2000:]
        assert pick(a.stderr) == pick(b.stderr) and len(pick(a.stderr)) >= 10
        if solver == "chain":
            # three strips, each reading only its own rows and columns of the file
            c3 = subprocess.run([ADMM, str(f), "--solver", solver, "--iters", "10", "--strips", "3"], capture_output=True, text=True, timeout=300)
            assert c3.returncode == 0, c3.stderr[-2000:]
            assert pick(c3.stderr) == pick(a.stderr)
            # ... and a strip of several that loaded from a file can save it again
            f2 = tmp_path / "saved_by_a_strip.nc"
            c4 = subprocess.run([ADMM, str(f), "--solver", solver, "--iters", "1", "--strips", "3", "--save", str(f2)],
                                capture_output=True, text=True, timeout=300)
            assert c4.returncode == 0, c4.stderr[-2000:]
            from admm_amd import graphfile as GF
            assert np.array_equal(GF.read_unaries(str(f2), negate=False), GF.read_unaries(str(f), negate=False))
            assert np.array_equal(GF.read_edges(str(f2)), GF.read_edges(str(f)))
    # a file that does not exist is an error message and a non-zero status
    c = subprocess.run([ADMM, str(tmp_path / "none.nc")], capture_output=True, text=True, timeout=60)
    assert c.returncode != 0 and "cannot open" in c.stderr
