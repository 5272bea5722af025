"""The C++ host (classes Grid / GraphBase / BBox with the reference's method names) and
its driver, which runs the reference's loop (admm.cpp:41-125) over the C ABI."""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADMM = os.path.join(ROOT, "admm_amd", "admm")
LABELS = ["processes", "allocTime", "loadTime", "initTime", "optTime", "maxCompTime", "maxCommTime",
          "maxUpdateTime", "minCompTime", "minCommTime", "minUpdateTime", "maxTotal"]


def _run(*args):
    assert os.path.exists(ADMM), "build the driver with `make -C admm_amd/csrc`"
    p = subprocess.run([ADMM] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stderr


def test_driver_reproduces_reference_objective_trace():
    """64 x 64, k = 4, Potts, 500 iterations: the reference's own plumbing configuration."""
    err = _run(64, 64, 4, "--solver", "edge")
    objs = [float(x) for x in re.findall(r"totalObj:\s+([-0-9.eE+]+)", err)]
    assert len(objs) == 500
    known = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))["64x64_k4"]
    for it, val in known.items():
        assert abs(objs[int(it) - 1] - val) <= 1e-5 * abs(val), (it, objs[int(it) - 1], val)
    # the 12 timing lines of admm.cpp:112-125, in order
    found = re.findall(r"^(\w+):\s+[-0-9.]+$", err, flags=re.M)
    tail = [x for x in found if x != "totalObj"]
    assert tail == LABELS


def test_driver_chain_strips_agree():
    a = _run(96, 80, 8, "--solver", "chain", "--iters", 20, "--pairwise", "trunc_linear")
    b = _run(96, 80, 8, "--solver", "chain", "--iters", 20, "--pairwise", "trunc_linear", "--strips", 4)
    pick = lambda s: re.findall(r"(totalObj|primal|dual):\s+([-0-9.]+)", s)
    assert pick(a) == pick(b)
    assert len(pick(a)) == 22


def test_driver_shows_a_tree_under_the_reference_member_names():
    """3 x 3, k = 2, three iterations: edge 0's marginals and multipliers as the unmodified
    reference printed them (class Edge members, Tree.h:26-35), through the host Edge view."""
    known = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))["3x3_k2"]
    err = _run(3, 3, 2, "--solver", "edge", "--iters", 3, "--tree", 0)
    row = lambda name: [float(x) for x in re.search(r"^%s:((?: [-0-9.eE+]+)+)$" % name, err, flags=re.M).group(1).split()]
    np = __import__("numpy")
    np.testing.assert_allclose(row("x_mu_node"), known["it3_edge0_mu"], rtol=1e-5)
    np.testing.assert_allclose(row("lambda_node"), known["it3_edge0_lambda"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(row("x_potential_node"), known["edge0_x_potential_node"], rtol=1e-7)
    assert len(row("x_mu_edge")) == 4 and len(row("edge_mu_node")) == 4


def test_driver_shows_a_chain():
    err = _run(6, 9, 4, "--solver", "chain", "--iters", 2, "--tree", 7, "--quiet")  # 6 row chains, then column 1
    m = re.search(r"^column chain 1, x:((?: \d+)+)$", err, flags=re.M)
    assert m and len(m.group(1).split()) == 6
