"""Clock stamps of the sweep's wavefronts (forward pass, trace-back) on one problem.

    python tools/gpu/stamps.py ROWS COLS K PAIRWISE [strips=G] [batch=B] [name=value ...]   (environment settings)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
R, C, k = map(int, sys.argv[1:4])
pairwise = sys.argv[4]
G = 1
B = 1
for kv in sys.argv[5:]:
    n, v = kv.split("=")
    if n == "strips":   # strip 0 of that many, alone
        G = int(v)
    elif n == "batch":  # that many independent grids
        B = int(v)
    else:
        os.environ[n] = v
import admm_amd as A  # noqa: E402
g = A.Grid(rank=0, nprocs=G)
g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k, batch=B)
g.Generate(seed=1, pairwise=pairwise); g.InitOptimization()
print(g.SweepKernelName(), sys.argv[5:], flush=True)
for _ in range(5):
    g.Compute(); g.Update()
g.Synchronize()
os.environ["ADMM_SWEEP_STAMPS"] = "1"
for _ in range(3):
    g.Compute(); g.Update()
g.Synchronize()
