#!/usr/bin/env python3
"""Per-GPU cost of one ADMM iteration when the 2048 x 2048, k = 32 grid is cut
into G strips: strip 0 of G is run alone on one GPU (Compute + Update, no
exchange partner, so the labels it would receive stay as they are), which is
what every GPU of a G-GPU job does between two exchanges.

    python tools/strip_timing.py [G ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_amd as A  # noqa: E402


def run(G, rows=2048, cols=2048, k=32, steps=50, warmup=5):
    g = A.Grid(rank=0, nprocs=G, device=0)
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(rows, cols, k)
    g.Generate(seed=1234, pairwise="potts")
    g.InitOptimization()
    g._unary = None
    for _ in range(warmup):
        g.Compute(); g.Update()
    g.Synchronize()
    g.ProfileReset(True)
    t0 = time.perf_counter()
    for i in range(steps):
        g.Compute(); g.Update(); g.ObjectiveEnqueue(1 + i)
    g.Synchronize()
    dt = time.perf_counter() - t0
    ms, n = g.ProfileRead()
    print("G=%d: %.3f ms/iteration (%.0f it/s), sweep %.3f ms over %d launches" % (G, 1e3 * dt / steps, steps / dt, ms, n), flush=True)
    g.close()


if __name__ == "__main__":
    for G in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
        run(G)
