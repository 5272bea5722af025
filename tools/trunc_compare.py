#!/usr/bin/env python3
"""Sweep time on a truncated linear table (16 labels, constant from distance 4) of the
kernel that knows the table's shape and of the general one (ADMM_SWEEP_TRUNC=1 / 0),
for several grid sizes.

    python tools/trunc_compare.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# rows, cols, k, batch, strips (strip 0 of that many is run alone)
CASES = [(128, 128, 16, 1, 1), (256, 256, 16, 1, 1), (384, 384, 16, 1, 1), (512, 512, 16, 1, 1), (1024, 1024, 16, 1, 1), (2048, 2048, 16, 1, 1), (4096, 4096, 16, 1, 1), (64, 64, 16, 256, 1)]

WORKER = r"""
import sys, time
sys.path.insert(0, %r)
import admm_amd as A
R, C, k, B, G = map(int, sys.argv[1:6])
g = A.Grid(rank=0, nprocs=G)
g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k, batch=B)
g.Generate(seed=1, pairwise="trunc_linear"); g.InitOptimization()
for _ in range(5):
    g.Compute(); g.Update()
g.Synchronize(); g.ProfileReset(True)
for _ in range(30):
    g.Compute(); g.Update()
g.Synchronize()
ms, n = g.ProfileRead()
print(g.SweepKernelName(), file=sys.stderr)
print("%%.4f" %% ms)
""" % ROOT

if __name__ == "__main__":
    for case in CASES:
        row = []
        for pc in ("1", "0"):
            env = dict(os.environ, ADMM_SWEEP_TRUNC=pc)
            out = subprocess.run([sys.executable, "-c", WORKER] + [str(v) for v in case], env=env, capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(out.stderr[-2000:])
                sys.exit(1)
            row.append(float(out.stdout.strip().splitlines()[-1]))
        print("%dx%d k=%d batch=%d, strip 0 of %d: table-shaped step %.4f ms, general step %.4f ms per sweep" % (case + tuple(row)), flush=True)
