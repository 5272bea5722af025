import sys, os
sys.path.insert(0, os.getcwd())
import admm_amd as A
for G in (8, 4):
    g = A.Grid(rank=0, nprocs=G, device=0)
    g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(2048, 2048, 32)
    g.Generate(seed=1234, pairwise="potts"); g.InitOptimization()
    for _ in range(3):
        g.Compute(); g.Update()
    g.Synchronize()
    print("NAME", G, g.SweepKernelName(), flush=True)
    g.close()
