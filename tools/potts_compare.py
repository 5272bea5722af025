#!/usr/bin/env python3
"""Sweep time of the kernel specialised for Potts tables against the general kernel
(ADMM_SWEEP_POTTS=0) on the same Potts problem, for several label counts.

    python tools/potts_compare.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(512, 512, 8, 1), (512, 512, 16, 1), (2048, 2048, 8, 1), (2048, 2048, 16, 1), (2048, 2048, 24, 1), (2048, 2048, 32, 1), (1024, 1024, 64, 1), (256, 256, 8, 256)]

WORKER = r"""
import sys, time
sys.path.insert(0, %r)
import admm_amd as A
R, C, k, B = map(int, sys.argv[1:5])
g = A.Grid()
g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(R, C, k, batch=B)
g.Generate(seed=1, pairwise="potts"); g.InitOptimization()
for _ in range(5):
    g.Compute(); g.Communicate(); g.Update()
g.Synchronize(); g.ProfileReset(True)
for _ in range(30):
    g.Compute(); g.Communicate(); g.Update()
g.Synchronize()
ms, n = g.ProfileRead()
print("%%.4f" %% ms)
""" % ROOT

if __name__ == "__main__":
    for case in CASES:
        row = []
        for potts in ("1", "0"):
            env = dict(os.environ, ADMM_SWEEP_POTTS=potts)
            out = subprocess.run([sys.executable, "-c", WORKER] + [str(v) for v in case], env=env, capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(out.stderr[-2000:])
                sys.exit(1)
            row.append(float(out.stdout.strip().splitlines()[-1]))
        print("%dx%d k=%d batch=%d: Potts kernel %.3f ms, general kernel %.3f ms per sweep" % (case + tuple(row)), flush=True)
