#!/usr/bin/env python3
"""Randomised parity of the edge solver against the oracle (GPU box): random grid shapes, label
counts, penalties, per-edge or shared tables, potentials of random scale (large ones drive
marginals into the subnormal range and to zero).  Prints the worst relative deviation.
    python tools/stress_edge.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_amd as A  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst = 0.0
    worst_obj = 0.0
    inexact = 0
    for case in range(cases):
        R, C = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        if R * C < 2:
            C = 2
        k = int(rng.choice([2, 3, 4, 5, 7, 8, 11, 16, 24, 32, 48, 64]))
        pen = float(rng.choice([0.5, 0.1, 1.0, 0.3, 2.0]))
        prime = float(rng.choice([1.0, 0.5, 0.3, 2.0, 1.7]))
        rho = float(rng.choice([1.0, 0.5, 2.0]))
        scale = float(rng.choice([1.0, 4.0, 30.0, 120.0]))
        un = (rng.random((R * C, k), dtype=np.float32) * scale).astype(np.float32)
        el = O.grid_edges(R, C)
        E = el.size // 2
        per_edge = bool(rng.integers(0, 2))
        if per_edge:
            ep = (rng.random(E * k * k, dtype=np.float32) * min(scale, 30.0)).astype(np.float32)
        else:
            ep = np.tile(O.pairwise_table("potts", k).reshape(-1), E)
        o = O.EdgeOracle(R * C, el, k, un, ep, penalty=pen, penalty_prime=prime, rho=rho)
        o.init()
        g = A.Grid()
        g.SetPartitioner(A.EDGE); g.SetPenalties(pen, prime); g.SetRhoEdge(rho)
        g.SetGlobalInfo(R, C, k)
        g.Load(un, ep if per_edge else ep[:k * k], per_edge=per_edge)
        g.InitOptimization()
        iters = int(rng.integers(1, 9))
        for it in range(iters):
            o.compute(); o.update()
            g.Compute(); g.Communicate(); g.Update()
        st = o.state()
        dev = 0.0
        for which, n in ((0, "mu_node"), (3, "lambda_node"), (4, "mu_edge"), (2, "m_node")):
            a, b = g.State(which).astype(np.float64), st[n].astype(np.float64).reshape(-1)
            if not (np.array_equal(np.isnan(a), np.isnan(b))):
                print("case %d: NaN pattern differs in %s" % (case, n)); sys.exit(1)
            ok = ~np.isnan(a)
            d = np.abs(a[ok] - b[ok]) / np.maximum(np.abs(b[ok]), 1e-7 if n != "mu_edge" else 1e-30)
            dev = max(dev, float(d.max()) if d.size else 0.0)
        inexact += dev != 0.0  # any binary32 value of the state that is not the oracle's
        oa, ob = o.objective(), g.ComputeLinearObj()
        if not (np.isnan(oa) and np.isnan(ob)):
            worst_obj = max(worst_obj, abs(oa - ob) / max(abs(oa), 1e-30))
            dev = max(dev, abs(oa - ob) / max(abs(oa), 1e-30))
        g.close()
        worst = max(worst, dev)
        if dev > 1e-5:
            print("case %d: %dx%d k=%d pen=%g prime=%g rho=%g scale=%g per_edge=%d iters=%d: deviation %.3g" % (case, R, C, k, pen, prime, rho, scale, per_edge, iters, dev))
            sys.exit(1)
        if case % 50 == 49:
            print("%d cases, worst relative deviation %.3g" % (case + 1, worst), flush=True)
    print("%d cases, worst relative deviation %.3g (objective, a binary64 sum in another order: %.3g); "
          "cases with any binary32 state value different from the oracle's: %d" % (cases, worst, worst_obj, inexact))


if __name__ == "__main__":
    main()
