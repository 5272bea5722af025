import os, sys, time
sys.path.insert(0, os.getcwd())
import admm_amd as A
g = A.Grid(); g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(2048, 2048, 32); g.Generate()
g.InitOptimization(); g.close()
g = A.Grid(); g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(2048, 2048, 32); g.Generate()
t = time.perf_counter(); g.InitOptimization(); g.Synchronize(); t1 = time.perf_counter() - t
t = time.perf_counter()
lab = g.Labels(); t2 = time.perf_counter() - t
print("create + upload 512 MiB + init: %.1f ms; label download 16 MiB: %.1f ms" % (1e3 * t1, 1e3 * t2))
