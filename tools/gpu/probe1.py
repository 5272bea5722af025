"""Probes: (1) how the multiplier counts grow over a 500-iteration run, (2) stamps."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import admm_amd as A

def counts(R, C, k, pw, iters):
    g = A.Grid(); g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k)
    g.Generate(seed=1234, pairwise=pw); g.InitOptimization()
    for it in range(1, iters + 1):
        g.Compute(); g.Communicate(); g.Update()
        if it in (1, 2, 5, 10, 20, 50, 100, 127, 200, 300, 400, 500):
            n = g.State(10) / 0.25
            a = np.abs(n)
            print("%dx%d k=%d %s it %d: max |n| %d, mean |n| %.3f, nonzero %.3f, >7: %.5f, >127: %d, labels changed along rows %.3f"
                  % (R, C, k, pw, it, a.max(), a.mean(), (a > 0).mean(), (a > 7).mean(), (a > 127).sum(),
                     (np.diff(g.State(13).reshape(R, C), axis=1) != 0).mean()), flush=True)
    g.close()

counts(512, 512, 32, "potts", 500)
counts(256, 256, 8, "potts", 300)
