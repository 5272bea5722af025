"""One strip on one device: the sweep reads the crossing chains' labels where those chains wrote
them (admm_update launches nothing) and queued objective sums ride along with the next sweep's
launch.  Both are checked against the forms they replace (ADMM_DIRECT_LABELS=0: Update
transposes the labels, which stays the choice where the sweep runs at the memory system's rate;
ADMM_DEFER_SUMS=0: every sum launched where it is queued) and against the oracle: labels and
multipliers bit-exact, objective sums bit-equal between the two forms (the same association) and
within the binary64 reordering tolerance against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OBJ_RTOL = 1e-9  # binary64 sums in a different order


def _grid(R, C, k, un, W, batch=1):
    import admm_amd as A
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k, batch=batch)
    g.Load(un, W)
    g.InitOptimization()
    return g


def _problem(R, C, k, pw, B, seed):
    from oracle import oracle as O
    W = O.pairwise_table(pw, k, 99) if pw == "dense" else O.pairwise_table(pw, k)
    un = np.concatenate([O.lcg_fill(seed + b, R * C * k)[0] for b in range(B)])
    return un, W


def _snapshot(g):
    return [g.State(13).copy(), g.State(14).copy(), g.Labels().copy(), g.State(10).copy(), g.State(11).copy(), g.Consensus().copy()]


CASES = [(8, 8, 4, "potts", 1), (33, 70, 8, "potts", 3), (40, 130, 16, "trunc_linear", 1), (70, 65, 32, "potts", 1),
         (66, 129, 32, "dense", 1), (20, 67, 64, "dense", 1), (9, 200, 5, "dense", 2), (1, 40, 4, "potts", 1),
         (40, 1, 4, "potts", 1), (130, 3, 64, "potts", 1), (64, 200, 16, "trunc_linear", 2), (300, 70, 32, "potts", 1)]


@pytest.mark.parametrize("R,C,k,pw,B", CASES)
def test_direct_labels_equal_transposed_labels(R, C, k, pw, B, monkeypatch):
    """Every iteration: the two forms of Update give the same labels, multipliers and objectives,
    and (one grid) the oracle's."""
    from oracle import oracle as O
    un, W = _problem(R, C, k, pw, B, 321)
    monkeypatch.setenv("ADMM_DIRECT_LABELS", "0")
    gt = _grid(R, C, k, un, W, B)
    monkeypatch.setenv("ADMM_DIRECT_LABELS", "1")
    gd = _grid(R, C, k, un, W, B)
    o = None
    if B == 1:
        o = O.ChainOracle(R, C, k, un, W)
        o.init()
    for it in range(6):
        for g in (gt, gd):
            g.Compute(); g.Communicate(); g.Update()
        st, sd = _snapshot(gt), _snapshot(gd)
        for a, b in zip(st, sd):
            assert np.array_equal(a, b), "iteration %d" % (it + 1)
        if o is not None:
            o.compute(); o.update()
            assert np.array_equal(sd[0].reshape(R, C), o.xrow) and np.array_equal(sd[1].reshape(R, C), o.xcol)
            assert np.array_equal(sd[2], o.labels())
        if it in (0, 3, 5):
            pt, dt = gt.Objective()
            pd, dd = gd.Objective()
            assert pt == pd and dt == dd
            assert gt.ComputeLinearObj() == gd.ComputeLinearObj()
            if o is not None:
                po, do, lo = o.objective()
                assert abs(pd - po) <= OBJ_RTOL * abs(po) and abs(dd - do) <= OBJ_RTOL * abs(do)
    gt.close(); gd.close()


@pytest.mark.parametrize("R,C,k,pw,B", [(33, 70, 8, "potts", 2), (70, 65, 32, "potts", 1), (30, 45, 16, "trunc_linear", 1),
                                        (20, 67, 64, "dense", 1)])
def test_direct_labels_any_call_order(R, C, k, pw, B, monkeypatch):
    """Compute without Update in between, Update twice, objectives and downloads between the
    calls: the crossing chains see the labels of the last Update, the chains' own labels are
    always the newest, in both forms of Update."""
    un, W = _problem(R, C, k, pw, B, 55)
    monkeypatch.setenv("ADMM_DIRECT_LABELS", "0")
    gt = _grid(R, C, k, un, W, B)
    monkeypatch.setenv("ADMM_DIRECT_LABELS", "1")
    gd = _grid(R, C, k, un, W, B)
    script = ["C", "U", "C", "C", "O", "U", "U", "C", "L", "C", "C", "C", "U", "O", "C", "U", "L", "C"]
    for n, op in enumerate(script):
        outs = []
        for g in (gt, gd):
            if op == "C":
                g.Compute()
            elif op == "U":
                g.Update()
            elif op == "O":
                outs.append(g.Objective())          # primal reads the transposed labels
            else:
                outs.append(g.ComputeLinearObj())
        if outs:
            assert outs[0] == outs[1], "call %d (%s)" % (n, op)
        for a, b in zip(_snapshot(gt), _snapshot(gd)):
            assert np.array_equal(a, b), "call %d (%s)" % (n, op)
    gt.close(); gd.close()


@pytest.mark.parametrize("R,C,k,pw,B", [(30, 45, 16, "trunc_linear", 1), (70, 65, 32, "potts", 1), (33, 70, 8, "potts", 3),
                                        (20, 67, 64, "dense", 1), (9, 200, 5, "dense", 1)])
def test_sums_in_the_next_sweeps_launch_equal_launched_sums(R, C, k, pw, B, monkeypatch):
    """Queued objectives whose sums ride along with the next sweep against the same run with every
    sum launched where it is queued: the same bits (one association), whatever comes between
    queueing and reading - a read before the next Compute, the same solve queued twice, a Compute
    with nothing queued, blocking calls in between."""
    un, W = _problem(R, C, k, pw, B, 9)

    def run():
        g = _grid(R, C, k, un, W, B)
        seen = []
        for it in range(9):
            g.Compute(); g.Communicate(); g.Update()
            if it != 6:                     # a solve nobody asks about
                g.ObjectiveEnqueue(1 + it)
            if it == 2:
                g.ObjectiveEnqueue(30)      # the same solve queued twice
            if it == 4:
                seen.append(g.ObjectiveRead(5, 1)[0])   # read while both terms are outstanding
            if it in (1, 5):
                seen.append(g.ComputeLinearObj())       # blocking call between queued ones
            if it == 7:
                g.Compute()                 # two sweeps, no Update, nothing queued in between
        seen.extend(g.ObjectiveRead(1, 6))
        seen.extend(g.ObjectiveRead(8, 2))
        seen.append(g.ObjectiveRead(30, 1)[0])
        blocking = g.ComputeLinearObj()
        g.close()
        return np.array(seen), blocking

    monkeypatch.setenv("ADMM_DEFER_SUMS", "0")
    a, ba = run()
    monkeypatch.delenv("ADMM_DEFER_SUMS")
    b, bb = run()
    assert np.array_equal(a, b) and ba == bb
    # slot 5 read early and read again; slots 2 and 6 against the blocking calls of the same
    # solves and slot 30 against slot 3 (unary terms summed by the gather in one, by the next
    # sweep in the other: another order)
    assert a[1] == a[3 + 4]
    for x, y in ((a[0], a[3 + 1]), (a[2], a[3 + 5]), (a[-1], a[3 + 2])):
        assert abs(x - y) <= OBJ_RTOL * abs(y)


def test_full_size_headline_both_forms(monkeypatch):
    """2048 x 2048, k = 32 Potts: ten iterations, labels of the two forms of Update bit-equal."""
    from oracle import oracle as O
    R = C = 2048
    k = 32
    un, W = O.synth_problem(R, C, k, 1234, "potts")
    res = []
    for env in ("0", "1"):
        monkeypatch.setenv("ADMM_DIRECT_LABELS", env)
        g = _grid(R, C, k, un, W)
        for _ in range(10):
            g.Compute(); g.Communicate(); g.Update()
        res.append((g.State(13).copy(), g.State(14).copy(), g.Labels().copy(), g.ComputeLinearObj()))
        g.close()
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b)
    assert res[0][3] == res[1][3]


@pytest.mark.parametrize("knob,value", [("ADMM_SWEEP_XCD", "1"), ("ADMM_SWEEP_XCD", "0"), ("ADMM_SWEEP_LDS_PAD", "40000")])
@pytest.mark.parametrize("R,C,k,pw,B", [(40, 40, 8, "potts", 20), (300, 70, 32, "potts", 1), (64, 200, 16, "trunc_linear", 2)])
def test_placement_knobs_change_no_result(knob, value, R, C, k, pw, B, monkeypatch):
    """Which die a workgroup's chains run on and how many workgroups a compute unit holds are
    tuning choices: the labels and the objective must not depend on them."""
    un, W = _problem(R, C, k, pw, B, 77)
    out = []
    for setting in (None, value):
        if setting is None:
            monkeypatch.delenv(knob, raising=False)
        else:
            monkeypatch.setenv(knob, setting)
        g = _grid(R, C, k, un, W, B)
        for _ in range(5):
            g.Compute(); g.Communicate(); g.Update()
        out.append(_snapshot(g) + [np.array([g.ComputeLinearObj()])])
        g.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)
