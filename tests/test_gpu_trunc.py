"""The sweep kernel specialised for truncated tables, W[a][b] = g(|a - b|) with g constant
from distance 4 on and nowhere above that constant (truncated linear, truncated quadratic,
Potts ...), against the oracle's general min-plus recursion and against the general kernel on
the same inputs (ADMM_SWEEP_TRUNC=1 / 0).  Bit-exact: labels of every iteration, both multiplier
arrays; objectives to 1e-9 (binary64 sums in another order).  With a penalty whose half is a
power of two the kernel keeps the multipliers as int8 counts (then int16, then binary32, as the
Potts kernel does); every form is run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _table(k, kind, s=0.25):
    d = np.abs(np.arange(k)[:, None] - np.arange(k)[None, :]).astype(np.float32)
    if kind == "linear":
        return (np.minimum(d, 4.0) * np.float32(s)).astype(np.float32)
    if kind == "quadratic":
        return (np.minimum(d * d, 10.0) * np.float32(s)).astype(np.float32)   # 0, 1, 4, 9, 10, 10 ...
    if kind == "short":                                                        # constant from distance 2 on
        return (np.minimum(d, 2.0) * np.float32(s)).astype(np.float32)
    if kind == "offset":                                                       # nonzero diagonal, irregular steps
        g = np.array([0.125, 0.5, 0.5, 0.875] + [1.0] * 16, dtype=np.float32)
        return g[d.astype(np.int64)]
    if kind == "infinite":                                                     # far labels forbidden
        g = np.array([0.0, 0.25, 0.5, 0.75] + [np.inf] * 16, dtype=np.float32)
        return g[d.astype(np.int64)]
    raise ValueError(kind)


def _run(R, C, k, un, W, G, iters, trunc_env, monkeypatch, penalty=0.5):
    import admm_amd as A
    if trunc_env is None:
        monkeypatch.delenv("ADMM_SWEEP_TRUNC", raising=False)
    else:
        monkeypatch.setenv("ADMM_SWEEP_TRUNC", trunc_env)
    gs = []
    for r in range(G):
        g = A.Grid(rank=r, nprocs=G)
        g.SetPartitioner(A.ROW)
        g.SetPenalties(penalty, 1.0)
        g.SetGlobalInfo(R, C, k)
        g.Load(un, W)
        g.InitOptimization()
        gs.append(g)
    names = [g.SweepKernelName() for g in gs]
    labels = []
    for _ in range(iters):
        for g in gs:
            g.Compute()
        if G > 1:
            A.communicate_group(gs)
        for g in gs:
            g.Update()
        labels.append(np.concatenate([g.Labels().reshape(-1, C) for g in gs if g.StripInfo()[1] > 0], axis=0))
    out = dict(
        names=names, labels=labels,
        lam_r=np.concatenate([g.State(10).reshape(-1, C, k) for g in gs if g.StripInfo()[1] > 0], axis=0),
        lam_c=np.concatenate([g.State(11).reshape(R, -1, k) for g in gs if g.StripInfo()[3] > 0], axis=1),
        xrow=np.concatenate([g.State(13).reshape(-1, C) for g in gs if g.StripInfo()[1] > 0], axis=0),
        xcol=np.concatenate([g.State(14).reshape(R, -1) for g in gs if g.StripInfo()[3] > 0], axis=1),
        obj=(sum(g.Objective()[0] for g in gs), sum(g.Objective()[1] for g in gs), sum(g.ComputeLinearObj() for g in gs)))
    for g in gs:
        g.close()
    return out


def _is_trunc(name):
    return name.split(",")[-1].strip(" >") == "3"


def _check(R, C, k, un, W, monkeypatch, G=1, iters=4, expect_trunc=True, penalty=0.5, counts=None):
    from oracle import oracle as O
    o = O.ChainOracle(R, C, k, un, W, penalty=penalty)
    o.init()
    ref = []
    for _ in range(iters):
        o.compute(); o.update()
        ref.append(o.labels().copy())
    a = _run(R, C, k, un, W, G, iters, "1", monkeypatch, penalty)   # (left to itself the library takes it from 384 wavefronts on)
    assert all(_is_trunc(n) == expect_trunc for n in a["names"]), a["names"]
    if counts is not None and expect_trunc:   # the kernel the run started with (field 6: counts)
        assert all((n.split(",")[6].strip() == "true") == counts for n in a["names"]), a["names"]
    b = _run(R, C, k, un, W, G, iters, "0", monkeypatch, penalty)
    assert not any(_is_trunc(n) for n in b["names"]), b["names"]
    runs = [("kernel for truncated tables", a), ("general kernel", b)]
    if expect_trunc:
        # up to 1024 wavefronts the kernel runs as two wavefronts per half chain (field 9 of the name); the other form too
        two = a["names"][0].split(",")[9].strip() == "true"
        monkeypatch.setenv("ADMM_SWEEP_TWOWAVE", "0" if two else "1")
        a2 = _run(R, C, k, un, W, G, iters, "1", monkeypatch, penalty)
        monkeypatch.delenv("ADMM_SWEEP_TWOWAVE")
        assert all(_is_trunc(n) and (n.split(",")[9].strip() == "true") != two for n in a2["names"]), a2["names"]
        runs.append(("kernel for truncated tables, %s wavefronts per half chain" % ("one" if two else "two"), a2))
    for name, r in runs:
        for it in range(iters):
            assert np.array_equal(r["labels"][it], ref[it]), "%s: labels, iteration %d" % (name, it + 1)
        assert np.array_equal(r["xrow"], o.xrow) and np.array_equal(r["xcol"], o.xcol), name
        assert np.array_equal(r["lam_r"], o.lambda_row), name + ": row multipliers"
        assert np.array_equal(r["lam_c"], o.lambda_col), name + ": column multipliers"
        for x, y in zip(r["obj"], o.objective()):
            assert x == y or abs(x - y) <= 1e-9 * max(abs(y), 1e-30) + 1e-12, (name, r["obj"], o.objective())


def _unaries(R, C, k, seed, quant=None):
    rng = np.random.default_rng(seed)
    un = rng.random((R * C, k), dtype=np.float32) * 3
    if quant:
        un = (np.floor(un * quant) / quant).astype(np.float32)  # many exact ties
    return un


@pytest.mark.parametrize("kind", ["linear", "quadratic", "short", "offset", "infinite"])
@pytest.mark.parametrize("k", [9, 13, 16])
def test_truncated_tables_match_oracle_and_general_kernel(kind, k, monkeypatch):
    R, C = 70, 131
    _check(R, C, k, _unaries(R, C, k, 3 + k), _table(k, kind), monkeypatch, G=1)
    _check(R, C, k, _unaries(R, C, k, 4 + k, quant=4), _table(k, kind), monkeypatch, G=3)   # ties in most steps


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (1, 90), (90, 1), (5, 9), (4, 257), (260, 6), (33, 64)])
def test_truncated_tables_on_ragged_shapes(shape, monkeypatch):
    R, C = shape
    k = 16
    _check(R, C, k, _unaries(R, C, k, 11 + R, quant=4), _table(k, "linear"), monkeypatch, G=2 if R > 1 and C > 1 else 1, iters=5)


@pytest.mark.parametrize("env", [{"ADMM_LAM8": "0"}, {"ADMM_LAM16": "0"}, {"ADMM_LAM8_THR": "1", "ADMM_LAM8_EVERY": "1"},
                                 {"ADMM_LAM8_THR": "2", "ADMM_LAM8_EVERY": "3"}, {"ADMM_LAM16_LIMIT": "3"},
                                 {"ADMM_LAM16_LIMIT": "5", "ADMM_LAM8_THR": "1", "ADMM_LAM8_EVERY": "4"}],
                         ids=["int16", "binary32", "int8-to-int16-at-once", "int8-to-int16", "counts-to-binary32", "int8-then-binary32"])
@pytest.mark.parametrize("G", [1, 3])
def test_every_form_of_the_multipliers(env, G, monkeypatch):
    for name, v in env.items():
        monkeypatch.setenv(name, v)
    R, C, k = 70, 150, 16
    _check(R, C, k, _unaries(R, C, k, 21, quant=4), _table(k, "linear"), monkeypatch, G=G, iters=9)
    _check(R, C, k, _unaries(R, C, k, 22), _table(k, "quadratic"), monkeypatch, G=G, iters=9, counts=env.get("ADMM_LAM16") != "0")


def test_penalty_that_is_no_power_of_two_keeps_binary32(monkeypatch):
    R, C, k = 60, 140, 16
    _check(R, C, k, _unaries(R, C, k, 23, quant=4), _table(k, "linear"), monkeypatch, G=2, iters=5, penalty=0.3, counts=False)


def test_all_ties(monkeypatch):
    R, C, k = 9, 80, 16
    _check(R, C, k, np.zeros((R * C, k), dtype=np.float32), _table(k, "linear"), monkeypatch)
    # g = 0 everywhere is also the Potts form, and that kernel comes first
    _check(R, C, k, np.zeros((R * C, k), dtype=np.float32), np.zeros((k, k), dtype=np.float32), monkeypatch, expect_trunc=False)


@pytest.mark.parametrize("what", ["hump", "asymmetric", "late"])
def test_tables_outside_the_form_take_the_general_kernel(what, monkeypatch):
    k = 16
    W = _table(k, "linear")
    if what == "hump":          # a near weight above the far one
        d = np.abs(np.arange(k)[:, None] - np.arange(k)[None, :])
        W = W.copy(); W[d == 2] = 1.5
    elif what == "asymmetric":
        W = W.copy(); W[3, 5] += 0.125
    else:                       # still growing at distance 5
        d = np.abs(np.arange(k)[:, None] - np.arange(k)[None, :]).astype(np.float32)
        W = (np.minimum(d, 6.0) * 0.25).astype(np.float32)
    R, C = 30, 70
    _check(R, C, k, _unaries(R, C, k, 5, quant=4), W, monkeypatch, expect_trunc=False)


@pytest.mark.parametrize("size,chosen", [(256, False), (512, True)])
def test_truncated_linear_at_full_size_and_the_default_choice(size, chosen, monkeypatch):
    """512 x 512 (the benchmark's truncated linear configuration) takes this kernel; 256 x 256 leaves
    most SIMDs without a wavefront and takes the general one unless told otherwise."""
    import admm_amd as A
    from oracle import oracle as O
    import os
    O.set_threads(min(16, os.cpu_count() or 1))
    R = C = size
    k = 16
    un, W = O.synth_problem(R, C, k, 1234, "trunc_linear")
    monkeypatch.delenv("ADMM_SWEEP_TRUNC", raising=False)
    g = A.Grid(); g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k); g.Load(un, W); g.InitOptimization()
    assert _is_trunc(g.SweepKernelName()) == chosen, g.SweepKernelName()
    g.close()
    monkeypatch.setenv("ADMM_SWEEP_TRUNC", "1")
    o = O.ChainOracle(R, C, k, un, W); o.init()
    g = A.Grid(); g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k); g.Load(un, W); g.InitOptimization()
    assert _is_trunc(g.SweepKernelName()), g.SweepKernelName()
    for _ in range(6):
        o.compute(); o.update(); g.Compute(); g.Communicate(); g.Update()
    assert np.array_equal(g.Labels(), o.labels())
    assert np.array_equal(g.State(10).reshape(R, C, k), o.lambda_row)
    g.close()
