#!/usr/bin/env python3
"""Clock stamps of the headline sweep's wavefronts (forward pass, trace-back) at several
iteration numbers of one run: what part of a launch depends on how far the solve has come.
    python tools/gpu/stamps_over_iterations.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import admm_amd as A  # noqa: E402
g = A.Grid()
g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(2048, 2048, 32)
g.Generate(seed=1234, pairwise="potts"); g.InitOptimization()
it = 0
for upto in (3, 12, 18, 30, 60, 150):
    while it < upto:
        g.Compute(); g.Update(); it += 1
    g.Synchronize()
    os.environ["ADMM_SWEEP_STAMPS"] = "1"
    sys.stderr.write("iteration %d:\n" % (it + 1)); sys.stderr.flush()
    g.Compute(); g.Update(); it += 1
    g.Synchronize()
    del os.environ["ADMM_SWEEP_STAMPS"]
g.close()
