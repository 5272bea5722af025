#!/usr/bin/env python3
"""Duration of the sweep launch as a function of the iteration number (HIP events around
every Compute, averaged over windows of ten iterations).
    python tools/gpu/launch_times.py [iterations] [workload]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import admm_amd as A  # noqa: E402
from bench import WORKLOADS  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rows, cols, k, batch, pairwise = WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "2048x2048_k32_potts"]
g = A.Grid()
g.SetPartitioner(A.ROW)
g.SetPenalties(0.5, 1.0)
g.SetGlobalInfo(rows, cols, k, batch=batch)
g.Generate(seed=1234, pairwise=pairwise)
g.InitOptimization()
out = []
for w in range(n // 10):
    g.ProfileReset(True)
    for i in range(10):
        g.Compute(); g.Communicate(); g.Update(); g.ObjectiveEnqueue(1 + 10 * w + i)
    ms, cnt = g.ProfileRead()
    out.append(ms)
print(" ".join("%.4f" % v for v in out))
g.close()
