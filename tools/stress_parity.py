#!/usr/bin/env python3
"""Randomised parity sweep on a GPU box: random grid shapes, label counts, pairwise
tables, strip counts and exchange modes, every iteration's chain labels and the final
multipliers bit-compared with the CPU oracle.

    python tools/stress_parity.py [cases] [seed] [first case to run]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_amd as A  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)


def one(rng, idx, skip=False):
    k = int(rng.choice([2, 3, 4, 5, 7, 8, 11, 16, 17, 24, 32, 33, 48, 64]))
    shape = rng.integers(0, 4)
    if shape == 0:
        R, C = int(rng.integers(1, 12)), int(rng.integers(1, 300))
    elif shape == 1:
        R, C = int(rng.integers(1, 300)), int(rng.integers(1, 12))
    else:
        R, C = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    G = int(rng.choice([1, 1, 2, 3, 4, 7]))
    exchange = int(rng.integers(0, 2)) if G > 1 else 0
    pw = str(rng.choice(["potts", "trunc_linear", "dense"]))
    quant = bool(rng.integers(0, 3) == 0)
    seed = int(rng.integers(1, 10**6))
    if skip:  # only the draws, so that a later case can be run on its own
        return None
    un, W = O.synth_problem(R, C, k, seed, pw)
    if quant:  # quantised costs: many exact ties
        un = (np.floor(un * 4) / 4).astype(np.float32)
        W = (np.floor(W * 4) / 4).astype(np.float32)
    ns = os.environ.get("ADMM_SWEEP_NS")
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    gs = []
    for r in range(G):
        g = A.Grid(rank=r, nprocs=G)
        g.exchange = exchange
        g.SetPartitioner(A.ROW)
        g.SetPenalties(0.5, 1.0)
        g.SetGlobalInfo(R, C, k)
        g.Load(un, W)
        g.InitOptimization()
        gs.append(g)
    tag = "case %d: %dx%d k=%d G=%d exchange=%d %s quant=%s NS=%s" % (idx, R, C, k, G, exchange, pw, quant, ns)
    for it in range(4):
        o.compute(); o.update()
        for g in gs:
            g.Compute()
        if G > 1:
            A.communicate_group(gs)
        for g in gs:
            g.Update()
        lab = np.concatenate([g.Labels().reshape(-1, C) for g in gs if g.StripInfo()[1] > 0], axis=0) if R else None
        assert np.array_equal(lab, o.labels()), tag + " labels, iteration %d" % (it + 1)
    lam = np.concatenate([g.State(10).reshape(-1, C, k) for g in gs if g.StripInfo()[1] > 0], axis=0)
    assert np.array_equal(lam, o.lambda_row), tag + " multipliers"
    lamc = np.concatenate([g.State(11).reshape(R, -1, k) for g in gs if g.StripInfo()[3] > 0], axis=1)
    assert np.array_equal(lamc, o.lambda_col) and not np.signbit(lamc[lamc == 0]).any(), tag + " column multipliers"
    p = sum(g.Objective()[0] for g in gs)
    d = sum(g.Objective()[1] for g in gs)
    lin = sum(g.ComputeLinearObj() for g in gs)
    po, do, lo = o.objective()
    for a, b, n in ((p, po, "primal"), (d, do, "dual"), (lin, lo, "linear")):
        assert abs(a - b) <= 1e-9 * max(abs(b), 1e-30) + 1e-12, tag + " %s %r %r" % (n, a, b)
    for g in gs:
        g.close()
    return tag


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    for i in range(cases):
        t = one(rng, i, skip=i < first)
        if i % 20 == 0 and t:
            print("ok", t, flush=True)
    print("%d random cases bit-exact" % cases)
