"""Edge solver: time of admm_compute per table entry (edge x k x k) for a range of label counts."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import admm_amd as A
from oracle import oracle as O

for (R, C, k) in ((512, 512, 4), (512, 512, 8), (512, 512, 16), (512, 512, 24), (512, 512, 32), (256, 256, 48), (256, 256, 64)):
    un, W = O.synth_problem(R, C, k, 1234, "potts")
    g = A.Grid()
    g.SetPartitioner(A.EDGE); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k); g.Load(un, W); g.InitOptimization()
    for _ in range(3):
        g.Compute(); g.Communicate(); g.Update()
    g.Synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        g.Compute()
    g.Synchronize()
    dt = (time.perf_counter() - t0) / n
    E = R * (C - 1) + C * (R - 1)
    print("%dx%d k=%d: %.3f ms per solve, %.2f ps per table entry" % (R, C, k, dt * 1e3, dt / (E * k * k) * 1e12), flush=True)
    g.close()
