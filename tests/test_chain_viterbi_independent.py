"""The chain subproblem of the oracle (oracle/admm_oracle.c: chain_minsum, swept from both
ends towards the middle) against an independent Viterbi written here in numpy, at the full
size of the benchmark configurations.

The independent solver is the textbook one-ended recursion
    M_0 = t_0,   M_{t+1}[b] = t_{t+1}[b] + min_a (M_t[a] + W[a][b]),
back-pointer = lowest a attaining the minimum, back-track from the lowest argmin of the last
message.  It associates the binary32 sums differently from the two-ended sweep, so the two
may return different, equally good labellings where paths tie to within rounding; the test
therefore checks
  * the chain minimum of both, on every chain, to the rounding bound of a binary32 recursion
    of T steps (T * 2^-23 relative; measured at full size: 1.2e-5 between the two sweeps on
    the 2048-node chains -- two binary32 recursions associated differently cannot agree to
    1e-6 at that length);
  * the energy of both labellings (evaluated in binary64 from the inputs) against the
    binary64 optimum, 1e-6 relative, on every chain;
  * the labels of both against the binary64 min-marginal argmin at every node whose second
    best min-marginal is further away than the rounding the binary32 sums can have
    accumulated, and that those nodes are the large majority.
Inputs are the adjusted unaries of the second ADMM iteration (multipliers and consensus
terms present), rebuilt here from the oracle's state with the shell formulas of
Tree.cpp:70 and :85-86.
"""
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

from oracle import oracle as O

EPS32 = float(np.finfo(np.float32).eps)


def _adjusted_unaries(o, k, penalty=0.5, rho=1.0):
    """t for the next Compute of every row chain [R, C, k] and column chain [C, R, k], in
    binary32 with the reference's operation order: lambda += penalty * (mu - m)
    (Tree.cpp:70), t = rho * (theta + lambda) - penalty * (m - mu) (Tree.cpp:85-86)."""
    pen, rh = np.float32(penalty), np.float32(rho)
    eye = np.eye(k, dtype=np.float32)
    m = o.consensus
    out = []
    for x, lam in ((o.xrow, o.lambda_row), (o.xcol, o.lambda_col)):
        mu = eye[x]
        la = (lam + pen * (mu - m)).astype(np.float32)
        out.append((rh * (o.theta + la) - pen * (m - mu)).astype(np.float32))
    return out[0], np.ascontiguousarray(out[1].transpose(1, 0, 2))


def _one_ended_viterbi_f32(th, W):
    """Independent solver, binary32: labels [n, T], minima [n]."""
    n, T, k = th.shape
    M = th[:, 0, :].copy()
    bp = np.empty((n, T, k), dtype=np.uint8)
    for t in range(1, T):
        c = M[:, :, None] + W[None, :, :]          # [n, a, b], one rounded add per candidate
        bp[:, t, :] = c.argmin(axis=1)             # lowest a on ties
        M = th[:, t, :] + c.min(axis=1)            # one more rounded add
    x = np.empty((n, T), dtype=np.int32)
    x[:, T - 1] = M.argmin(axis=1)
    rows = np.arange(n)
    for t in range(T - 1, 0, -1):
        x[:, t - 1] = bp[rows, t, x[:, t]]
    return x, M.min(axis=1)


def _min_marginals_f64(th, W):
    """Binary64 min-marginals mm[n, T, k] = best energy of a labelling with node t at label a."""
    th = th.astype(np.float64)
    W = W.astype(np.float64)
    n, T, k = th.shape
    F = np.empty_like(th)
    B = np.empty_like(th)
    F[:, 0] = th[:, 0]
    for t in range(1, T):
        F[:, t] = th[:, t] + (F[:, t - 1, :, None] + W[None]).min(axis=1)
    B[:, T - 1] = 0.0
    for t in range(T - 2, -1, -1):
        B[:, t] = (W[None] + (th[:, t + 1] + B[:, t + 1])[:, None, :]).min(axis=2)
    return F + B


def _energy_f64(th, W, x):
    n, T, _ = th.shape
    e = np.take_along_axis(th, x[:, :, None].astype(np.int64), axis=2)[:, :, 0].astype(np.float64).sum(axis=1)
    if T > 1:
        e += W[x[:, :-1], x[:, 1:]].astype(np.float64).sum(axis=1)
    return e


def _check_block(args):
    th, W = args
    n, T, k = th.shape
    x2, v2 = O.chain_minsum_batch(th, W)            # the oracle: two-ended
    x1, v1 = _one_ended_viterbi_f32(th, W)          # independent: one-ended
    mm = _min_marginals_f64(th, W)
    opt = mm[:, 0, :].min(axis=1)
    scale = np.maximum(np.abs(opt), 1.0)
    rel12 = float((np.abs(v1.astype(np.float64) - v2) / scale).max())
    rel2o = float((np.abs(v2 - opt) / scale).max())
    # every step rounds twice, by at most half an ulp of a partial sum that never exceeds the
    # result (costs are non-negative): T * eps bounds either recursion against the exact value
    vtol = max(T, 2) * EPS32
    assert rel12 <= vtol, "chain minima of the two sweeps differ: %g relative, %g allowed" % (rel12, vtol)
    assert rel2o <= vtol, "binary32 chain minimum far from the binary64 optimum: %g relative" % rel2o
    for x in (x1, x2):
        assert x.min() >= 0 and x.max() < k
        assert np.all(np.abs(_energy_f64(th, W, x) - opt) <= 1e-6 * scale), "a labelling is not optimal"
    # nodes whose best label is unambiguous beyond accumulated binary32 rounding
    part = np.partition(mm, 1, axis=2)
    gap = part[:, :, 1] - part[:, :, 0]
    tol = (8.0 * np.sqrt(T) * EPS32 * scale)[:, None]
    clear = gap > tol
    best = mm.argmin(axis=2)
    assert np.array_equal(x1[clear], best[clear]), "one-ended labels differ from the min-marginal argmin"
    assert np.array_equal(x2[clear], best[clear]), "two-ended labels differ from the min-marginal argmin"
    return int(clear.sum()), int(clear.size), int((x1 != x2).sum()), rel12, rel2o


@pytest.mark.parametrize("R,C,k,pw", [(512, 512, 16, "trunc_linear"), (2048, 2048, 32, "potts"), (1024, 1024, 64, "dense")],
                         ids=["512x512_k16_trunc_linear", "2048x2048_k32_potts", "1024x1024_k64_dense"])
def test_two_ended_sweep_is_a_viterbi_at_full_size(R, C, k, pw):
    O.set_threads(min(8, os.cpu_count() or 1))
    un, W = O.synth_problem(R, C, k, 1234, pw)
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    o.compute(); o.update()                         # state of a running solve, not the first iteration's zeros
    th_row, th_col = _adjusted_unaries(o, k)
    del o
    O.set_threads(1)                                # the pool below is the parallelism
    blocks = []
    per = 128 if k >= 64 else 256
    for th in (th_row, th_col):
        for i in range(0, th.shape[0], per):
            blocks.append((th[i:i + per], W))
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(_check_block, blocks))
    clear = sum(r[0] for r in res)
    total = sum(r[1] for r in res)
    assert total == 2 * R * C
    assert clear >= 0.95 * total, "only %d of %d nodes are unambiguous: the label check is too weak" % (clear, total)
    print("%dx%d k=%d %s: %d of %d nodes unambiguous, %d labels differ between the sweeps, minima differ by at most %.2e relative "
          "(two-ended vs binary64 optimum: %.2e)" % (R, C, k, pw, clear, total, sum(r[2] for r in res), max(r[3] for r in res), max(r[4] for r in res)))


def test_two_ended_sweep_on_ragged_and_tiny_chains():
    rng = np.random.default_rng(5)
    for k, T, n in ((2, 1, 7), (3, 2, 9), (5, 3, 11), (7, 64, 5), (33, 65, 4), (64, 129, 3), (16, 1000, 2)):
        th = rng.random((n, T, k), dtype=np.float32)
        W = rng.random((k, k), dtype=np.float32)
        _check_block((th, W))
