"""Per-iteration time of the chain solver with and without the per-launch profiling events
(experiment; prints one line per workload and setting)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import admm_amd as A

W = {"2048": (2048, 2048, 32, 1, "potts"), "512": (512, 512, 16, 1, "trunc_linear"), "64": (64, 64, 4, 1, "potts"),
     "batch": (256, 256, 8, 256, "potts"), "dense": (1024, 1024, 64, 1, "dense"), "dense32": (2048, 2048, 32, 1, "dense"),
     "dense16": (2048, 2048, 16, 1, "dense"), "trunc32": (2048, 2048, 32, 1, "trunc_linear")}
names = sys.argv[1].split(",")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for name in names:
    rows, cols, k, batch, pw = W[name]
    g = A.Grid()
    g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(rows, cols, k, batch=batch)
    g.Generate(seed=1234, pairwise=pw); g.InitOptimization(); g._unary = None
    for prof in (True, False):
        for i in range(20):
            g.Compute(); g.Communicate(); g.Update(); g.ObjectiveEnqueue(1 + i)
        g.Synchronize()
        g.ProfileReset(prof)
        t0 = time.perf_counter()
        for i in range(steps):
            g.Compute(); g.Communicate(); g.Update(); g.ObjectiveEnqueue(30 + i)
        g.Synchronize()
        dt = time.perf_counter() - t0
        km = g.ProfileRead()[0] if prof else float("nan")
        g.ProfileReset(False)
        print("%s events=%d: %.4f ms per iteration (%.1f it/s), kernel %.4f ms" % (name, prof, 1e3 * dt / steps, steps / dt, km), flush=True)
    g.close()
