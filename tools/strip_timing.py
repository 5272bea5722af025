#!/usr/bin/env python3
"""Per-GPU cost of one ADMM iteration when the 2048 x 2048, k = 32 grid is cut
into G strips: strip 0 of G is run alone on one GPU (Compute + Update, no
exchange partner, so the labels it would receive stay as they are), which is
what every GPU of a G-GPU job does between two exchanges.

    python tools/strip_timing.py [G ...]
    python tools/strip_timing.py --group [G ...]

--group: all G strips in this process on one GPU, with the in-process exchange between
them (admm_communicate_group), so that an iteration also pays for moving the labels and
laying them out along the crossing chains.  The strips' launches share the one GPU, so
the time of an iteration is the SUM over the strips of everything a strip launches;
divided by G it is what one strip's GPU does per iteration, the wire time excepted.
ADMM_GROUP_PULL=0 times the pack / copy / transpose form of the exchange.
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


def run_group(G, rows=2048, cols=2048, k=32, steps=50, warmup=5):
    gs = []
    for r in range(G):
        g = A.Grid(rank=r, nprocs=G, device=0)
        g.SetPartitioner(A.ROW)
        g.SetPenalties(0.5, 1.0)
        g.SetGlobalInfo(rows, cols, k)
        g.Generate(seed=1234, pairwise="potts")
        g.InitOptimization()
        g._unary = None
        gs.append(g)

    def it(i):
        for g in gs:
            g.Compute()
        if G > 1:
            A.communicate_group(gs)
        for g in gs:
            g.Update()
            g.ObjectiveEnqueue(1 + i)

    for i in range(warmup):
        it(i)
    for g in gs:
        g.Synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        it(warmup + i)
    for g in gs:
        g.Synchronize()
    dt = time.perf_counter() - t0
    print("G=%d in one process: %.3f ms/iteration for all strips, %.3f ms per strip" % (G, 1e3 * dt / steps, 1e3 * dt / steps / G), flush=True)
    for g in gs:
        g.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    group = "--group" in args
    args = [a for a in args if a != "--group"]
    for G in [int(a) for a in args] or [1, 2, 4, 8]:
        (run_group if group else run)(G)
