"""The sweep kernel specialised for Potts tables (zero diagonal, one non-negative value
everywhere else) against the oracle's general min-plus recursion, and against the general
kernel on the same inputs (ADMM_SWEEP_POTTS=0 keeps the general kernel on a Potts table).
Bit-exact: labels of every iteration, both multiplier arrays, the three objectives."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _potts(k, w):
    W = np.full((k, k), w, dtype=np.float32)
    np.fill_diagonal(W, 0.0)
    return W


def _run_gpu(R, C, k, un, W, G, iters, potts_env, penalty=0.5):
    import admm_amd as A
    old = os.environ.get("ADMM_SWEEP_POTTS")
    if potts_env is None:
        os.environ.pop("ADMM_SWEEP_POTTS", None)
    else:
        os.environ["ADMM_SWEEP_POTTS"] = potts_env
    try:
        gs = []
        for r in range(G):
            g = A.Grid(rank=r, nprocs=G)
            g.SetPartitioner(A.ROW)
            g.SetPenalties(penalty, 1.0)
            g.SetGlobalInfo(R, C, k)
            g.Load(un, W)
            g.InitOptimization()
            gs.append(g)
        info = [g.SweepKernelInfo() for g in gs]
        labels = []
        for _ in range(iters):
            for g in gs:
                g.Compute()
            if G > 1:
                A.communicate_group(gs)
            for g in gs:
                g.Update()
            labels.append(np.concatenate([g.Labels().reshape(-1, C) for g in gs if g.StripInfo()[1] > 0], axis=0))
        lam_r = np.concatenate([g.State(10).reshape(-1, C, k) for g in gs if g.StripInfo()[1] > 0], axis=0)
        lam_c = np.concatenate([g.State(11).reshape(R, -1, k) for g in gs if g.StripInfo()[3] > 0], axis=1)
        xrow = np.concatenate([g.State(13).reshape(-1, C) for g in gs if g.StripInfo()[1] > 0], axis=0)
        xcol = np.concatenate([g.State(14).reshape(R, -1) for g in gs if g.StripInfo()[3] > 0], axis=1)
        obj = (sum(g.Objective()[0] for g in gs), sum(g.Objective()[1] for g in gs), sum(g.ComputeLinearObj() for g in gs))
        for g in gs:
            g.close()
        return dict(info=info, labels=labels, lam_r=lam_r, lam_c=lam_c, xrow=xrow, xcol=xcol, obj=obj)
    finally:
        if old is None:
            os.environ.pop("ADMM_SWEEP_POTTS", None)
        else:
            os.environ["ADMM_SWEEP_POTTS"] = old


def _check(R, C, k, un, W, G=1, iters=4, expect_potts=True, penalty=0.5):
    from oracle import oracle as O
    o = O.ChainOracle(R, C, k, un, W, penalty=penalty)
    o.init()
    ref_labels = []
    for _ in range(iters):
        o.compute(); o.update()
        ref_labels.append(o.labels().copy())
    a = _run_gpu(R, C, k, un, W, G, iters, None, penalty)
    assert all(i[2] == expect_potts for i in a["info"]), a["info"]
    b = _run_gpu(R, C, k, un, W, G, iters, "0", penalty)
    assert not any(i[2] for i in b["info"]), b["info"]
    for name, r in (("potts kernel" if expect_potts else "default kernel", a), ("general kernel", b)):
        for it in range(iters):
            assert np.array_equal(r["labels"][it], ref_labels[it]), "%s: labels, iteration %d" % (name, it + 1)
        assert np.array_equal(r["xrow"], o.xrow) and np.array_equal(r["xcol"], o.xcol), name
        assert np.array_equal(r["lam_r"], o.lambda_row), name + ": row multipliers"
        assert np.array_equal(r["lam_c"], o.lambda_col), name + ": column multipliers"
        assert not np.signbit(r["lam_c"][r["lam_c"] == 0]).any(), name + ": a zero multiplier came back as -0"
        po, do, lo = o.objective()
        for x, y in zip(r["obj"], (po, do, lo)):
            # (an infinite weight gives an infinite primal energy once the two copies disagree)
            assert x == y or abs(x - y) <= 1e-9 * max(abs(y), 1e-30) + 1e-12, (name, r["obj"], (po, do, lo))
    assert a["obj"] == b["obj"]


def _unaries(R, C, k, seed, quant=None):
    rng = np.random.default_rng(seed)
    un = rng.random((R * C, k), dtype=np.float32) * 4
    if quant:
        un = (np.floor(un * quant) / quant).astype(np.float32)  # many exact ties
    return un


@pytest.mark.parametrize("k", [8, 21, 32, 33, 64])
@pytest.mark.parametrize("G", [1, 3])
def test_potts_kernel_matches_oracle_and_general_kernel(k, G):
    R, C = 150, 262  # half chains of 75 and 131 steps: whole 64-step chunks and tails, odd and even
    _check(R, C, k, _unaries(R, C, k, 5 + k), _potts(k, 0.75), G=G)


@pytest.mark.parametrize("k", [5, 8, 13, 16])
@pytest.mark.parametrize("shape", [(150, 262), (1, 300), (300, 1), (700, 40)])
def test_potts_kernel_with_several_chains_per_wavefront(k, shape):
    # 8 or 16 label lanes: four or eight chains share a wavefront and the trace-back runs one
    # lane per chain (quantised unaries: ties in most steps)
    R, C = shape
    _check(R, C, k, _unaries(R, C, k, 11 + k, quant=4), _potts(k, 0.5), G=2 if R > 1 and C > 1 else 1)


def test_batch_of_grids_on_the_potts_kernel():
    import admm_amd as A
    from oracle import oracle as O
    R, C, k, B = 70, 90, 8, 5
    W = _potts(k, 0.5)
    uns = [_unaries(R, C, k, 40 + b, quant=8) for b in range(B)]
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k, batch=B)
    g.Load(np.concatenate(uns, axis=0), W)
    g.InitOptimization()
    assert g.SweepKernelInfo()[2]
    for _ in range(4):
        g.Compute(); g.Communicate(); g.Update()
    lab = g.Labels().reshape(B, R, C)
    g.close()
    for b in range(B):
        o = O.ChainOracle(R, C, k, uns[b], W)
        o.init()
        for _ in range(4):
            o.compute(); o.update()
        assert np.array_equal(lab[b], o.labels().reshape(R, C)), "grid %d" % b


@pytest.mark.parametrize("penalty", [0.5, 0.3, 4.0, 2.0 ** -20, 1.5])
@pytest.mark.parametrize("k", [8, 32])
def test_multipliers_as_counts_and_as_binary32(penalty, k):
    # One strip + Potts kernel + penalty / 2 a power of two: the multiplier arrays hold int16
    # counts.  Any other penalty keeps binary32.  Both against the oracle, bit for bit.
    R, C = 90, 200
    _check(R, C, k, _unaries(R, C, k, 21 + k, quant=4), _potts(k, 0.5), iters=6, penalty=penalty)
    _check(R, C, k, _unaries(R, C, k, 22 + k, quant=4), _potts(k, 0.5), G=2, iters=4, penalty=penalty)


@pytest.mark.parametrize("limit", ["0", "1", "3"])
@pytest.mark.parametrize("G", [1, 3])
def test_counts_turn_into_binary32_in_mid_run(limit, G, monkeypatch):
    # ADMM_LAM16_LIMIT: the switch that would come after 32000 sweeps, made to come after 0, 1, 3
    monkeypatch.setenv("ADMM_LAM16_LIMIT", limit)
    R, C, k = 70, 150, 32
    _check(R, C, k, _unaries(R, C, k, 77, quant=4), _potts(k, 0.5), G=G, iters=6)


def test_binary32_multipliers_on_request(monkeypatch):
    monkeypatch.setenv("ADMM_LAM16", "0")
    R, C, k = 70, 150, 16
    _check(R, C, k, _unaries(R, C, k, 78, quant=4), _potts(k, 0.5), iters=5)


def test_objectives_of_a_large_batch():
    # 66560 per-half-chain terms: above the size from which the objective sums run in two stages
    import admm_amd as A
    from oracle import oracle as O
    R, C, k, B = 128, 128, 8, 130
    W = _potts(k, 0.5)
    rng = np.random.default_rng(123)
    un = rng.random((B * R * C, k), dtype=np.float32)
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k, batch=B)
    g.Load(un, W)
    g.InitOptimization()
    queued = []
    for it in range(3):
        g.Compute(); g.Communicate(); g.Update()
        g.ObjectiveEnqueue(1 + it)
    g.Synchronize()
    queued = g.ObjectiveRead(1, 3)
    lin = g.ComputeLinearObj()
    primal, dual = g.Objective(dual=True)
    g.close()
    ref = np.zeros((3, 3))
    for b in range(B):
        o = O.ChainOracle(R, C, k, un[b * R * C:(b + 1) * R * C], W)
        o.init()
        for it in range(3):
            o.compute(); o.update()
            ref[it] += np.array(o.objective())
    for it in range(3):
        assert abs(queued[it] - ref[it][2]) <= 1e-9 * abs(ref[it][2]), (it, queued[it], ref[it][2])
    for x, y in ((primal, ref[2][0]), (dual, ref[2][1]), (lin, ref[2][2])):
        assert abs(x - y) <= 1e-9 * abs(y), (x, y)


@pytest.mark.parametrize("w", [0.0, 0.25, 1e-30, 1e30, float("inf")])
def test_potts_weights_with_ties(w):
    # quantised unaries and a weight on the same lattice (or zero, or absorbing): ties between
    # keeping the label and switching, and between several switch targets, in most steps
    R, C, k = 70, 200, 32
    _check(R, C, k, _unaries(R, C, k, 3, quant=4), _potts(k, w))


def test_constant_unaries_all_ties():
    R, C, k = 40, 140, 32
    un = np.zeros((R * C, k), dtype=np.float32)
    _check(R, C, k, un, _potts(k, 0.5))
    _check(R, C, k, un, _potts(k, 0.0))


def test_tables_that_are_not_potts_take_the_general_kernel():
    R, C, k = 33, 150, 32
    un = _unaries(R, C, k, 9, quant=8)
    W = _potts(k, 0.5); W[3, 7] = 0.25          # one entry off
    _check(R, C, k, un, W, expect_potts=False)
    W = _potts(k, -0.5)                           # negative weight: the candidate a = b can win
    _check(R, C, k, un, W, expect_potts=False)
    W = _potts(k, 0.5); W[5, 5] = 0.125           # non-zero diagonal
    _check(R, C, k, un, W, expect_potts=False)


def test_single_row_and_single_column():
    for R, C in ((1, 300), (300, 1), (2, 129), (1, 1)):
        k = 32
        _check(R, C, k, _unaries(R, C, k, R + C, quant=2), _potts(k, 0.5), iters=3)


@pytest.mark.parametrize("k", [21, 32, 64])
@pytest.mark.parametrize("shape", [(3, 2), (5, 15), (6, 17), (4, 127), (3, 130), (2, 513), (3, 1023), (2, 1026), (3, 2050), (2300, 3), (40, 33)])
def test_path_compression_in_the_trace_back(k, shape, monkeypatch):
    """The Potts kernel that records, per block of eight steps, where every label's path
    comes from, and traces back block-wise (chosen by itself only for long chains on a busy
    GPU; ADMM_SWEEP_PCOMP=1 forces it): half chains of 1 to 1150 steps, so partial blocks,
    partial 64-step chunks, partial rounds and several 512-step pieces all occur; quantised
    unaries tie in most steps."""
    monkeypatch.setenv("ADMM_SWEEP_PCOMP", "1")
    monkeypatch.setenv("ADMM_SWEEP_TWOWAVE", "0")
    R, C = shape
    _check(R, C, k, _unaries(R, C, k, 3 * k + R, quant=4), _potts(k, 0.5), G=1, iters=3)


@pytest.mark.parametrize("penalty,w", [(0.5, 0.0), (0.3, 0.75), (4.0, float("inf")), (0.5, 1e30)])
def test_path_compression_with_other_weights_and_penalties(penalty, w, monkeypatch):
    monkeypatch.setenv("ADMM_SWEEP_PCOMP", "1")
    monkeypatch.setenv("ADMM_SWEEP_TWOWAVE", "0")
    for k, (R, C), G in ((32, (70, 300), 3), (64, (33, 140), 2)):
        _check(R, C, k, _unaries(R, C, k, 17, quant=2), _potts(k, w), G=G, iters=4, penalty=penalty)


@pytest.mark.parametrize("G", [1, 3])
@pytest.mark.parametrize("k", [8, 32, 64])
def test_int16_counts_on_request(k, G, monkeypatch):
    """ADMM_LAM8=0: the counts start as int16 (the form they otherwise take once a count nears the int8 range)."""
    monkeypatch.setenv("ADMM_LAM8", "0")
    R, C = 70, 150
    _check(R, C, k, _unaries(R, C, k, 31 + k, quant=4), _potts(k, 0.5), G=G, iters=5)


@pytest.mark.parametrize("thr,every", [("1", "1"), ("2", "1"), ("1", "3"), ("2", "2")])
@pytest.mark.parametrize("G", [1, 3])
def test_int8_counts_turn_into_int16_in_mid_run(thr, every, G, monkeypatch):
    """The scan of the int8 counts reports |n| >= 1 or 2 right away, so the arrays change
    to int16 after a few sweeps, at whatever sweep the host happens to see the report."""
    monkeypatch.setenv("ADMM_LAM8_THR", thr)
    monkeypatch.setenv("ADMM_LAM8_EVERY", every)
    R, C, k = 70, 150, 32
    _check(R, C, k, _unaries(R, C, k, 79, quant=4), _potts(k, 0.5), G=G, iters=9)


@pytest.mark.parametrize("limit", ["2", "5"])
def test_int8_counts_then_binary32(limit, monkeypatch):
    """int8 -> (int16 ->) binary32 when the sweep limit of the counts comes first."""
    monkeypatch.setenv("ADMM_LAM16_LIMIT", limit)
    monkeypatch.setenv("ADMM_LAM8_THR", "1")
    monkeypatch.setenv("ADMM_LAM8_EVERY", "4")
    R, C, k = 40, 90, 16
    _check(R, C, k, _unaries(R, C, k, 80, quant=4), _potts(k, 0.5), G=2, iters=8)


@pytest.mark.parametrize("k,G", [(8, 1), (32, 1), (32, 2)])
def test_counts_that_outgrow_int8(k, G):
    """A tiny penalty leaves the two copies of many nodes in disagreement for hundreds of
    sweeps, so their counts grow by one per sweep and pass 127: with the default scan
    (threshold 64, every 16 sweeps) the arrays must have become int16 in time.  150
    iterations against the oracle, multipliers compared as binary32."""
    from oracle import oracle as O
    import admm_amd as A
    R, C = 12, 70
    pen = 2.0 ** -20
    un = _unaries(R, C, k, 5, quant=None)
    W = _potts(k, 0.05)
    o = O.ChainOracle(R, C, k, un, W, penalty=pen)
    o.init()
    gs = []
    for r in range(G):
        g = A.Grid(rank=r, nprocs=G)
        g.SetPartitioner(A.ROW); g.SetPenalties(pen, 1.0); g.SetGlobalInfo(R, C, k); g.Load(un, W); g.InitOptimization()
        gs.append(g)
    on_int8 = lambda g: g.SweepKernelName().split(",")[8].strip(" >") == "true"
    assert on_int8(gs[0]), gs[0].SweepKernelName()  # starts on int8 counts
    for it in range(150):
        o.compute(); o.update()
        for g in gs:
            g.Compute()
        if G > 1:
            A.communicate_group(gs)
        for g in gs:
            g.Update()
        if it in (60, 100, 126, 127, 128, 149):
            lam = np.concatenate([g.State(10).reshape(-1, C, k) for g in gs], axis=0)
            assert np.array_equal(lam, o.lambda_row), "row multipliers after %d sweeps" % (it + 1)
            lab = np.concatenate([g.Labels().reshape(-1, C) for g in gs], axis=0)
            assert np.array_equal(lab, o.labels()), "labels after %d sweeps" % (it + 1)
    n = np.abs(o.lambda_row / np.float32(pen / 2)).max()
    assert n > 127, "the case does not leave the int8 range (max count %d)" % n
    assert not on_int8(gs[0]), "still on int8 counts"
    lamc = np.concatenate([g.State(11).reshape(R, -1, k) for g in gs], axis=1)
    assert np.array_equal(lamc, o.lambda_col)
    for g in gs:
        g.close()


@pytest.mark.parametrize("two", ["0", "1"])
@pytest.mark.parametrize("penalty,w", [(0.5, 0.5), (0.3, 0.75), (4.0, float("inf")), (0.5, 0.0)])
@pytest.mark.parametrize("shape,G", [((150, 262), 1), ((150, 262), 3), ((3, 2), 1), ((1, 77), 1), ((70, 1), 1), ((5, 1030), 1), ((2050, 3), 2)])
def test_one_and_two_wavefronts_per_half_chain(two, penalty, w, shape, G, monkeypatch):
    """32 labels have a kernel that gives every half chain two wavefronts (one for the ADMM
    shell, one for the recursion; chosen by itself when chains are scarce) next to the one
    that gives it one: both forced here, on binary32 multipliers and on counts, whole and
    partial blocks, chunks and chains of one node."""
    monkeypatch.setenv("ADMM_SWEEP_TWOWAVE", two)
    R, C = shape
    _check(R, C, 32, _unaries(R, C, 32, 41 + R, quant=4), _potts(32, w), G=G, iters=4, penalty=penalty)


def test_two_wavefronts_with_int16_counts_and_21_labels(monkeypatch):
    monkeypatch.setenv("ADMM_SWEEP_TWOWAVE", "1")
    monkeypatch.setenv("ADMM_LAM8", "0")
    _check(90, 200, 21, _unaries(90, 200, 21, 9, quant=4), _potts(21, 0.5), G=2, iters=5)


@pytest.mark.parametrize("on", ["0", "1"])
@pytest.mark.parametrize("R,C", [(150, 262), (75, 40), (33, 130)])
def test_column_wavefront_pairs_on_one_die(R, C, on, monkeypatch):
    """ADMM_SWEEP_PAIRXCD: the pairs of column wavefronts that share cache lines of the int8
    counts are handed to workgroups eight apart (a permutation of every 16 mixed workgroups, the
    rest in place).  Which workgroup sweeps which chains must not show in any result: more,
    exactly and fewer than 16 mixed workgroups, forced on and off."""
    import admm_amd as A
    monkeypatch.setenv("ADMM_SWEEP_PAIRXCD", on)
    monkeypatch.setenv("ADMM_SWEEP_TWOWAVE", "0")  # (a problem this small would otherwise run the two-wavefront kernels, whose workgroups are not mixed)
    k = 32
    _check(R, C, k, _unaries(R, C, k, 17, quant=4), _potts(k, 0.5), G=1, iters=4)
    g = A.Grid()
    g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k)
    g.Load(_unaries(R, C, k, 17, quant=4), _potts(k, 0.5)); g.InitOptimization()
    g.Compute()
    info = g.LaunchInfo()
    assert info["pair_xcd"] == int(on) and info["mixed_workgroups"] == 1 and info["multipliers"] == "int8", info
    g.close()
