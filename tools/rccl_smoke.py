import os, sys
sys.path.insert(0, os.getcwd())
import admm_amd as A
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
import torch.distributed as dist
dist.init_process_group("gloo", rank=0, world_size=1)
g = A.Grid(rank=0, nprocs=1, device=0)
g.SetPartitioner(A.ROW); g.SetPenalties(0.5, 1.0); g.SetGlobalInfo(64, 64, 32); g.Generate(); g.InitOptimization()
ids = [A.unique_id()]
dist.broadcast_object_list(ids, src=0)
g.CommInit(ids[0])
for i in range(3):
    g.Compute(); g.Communicate(); g.Update()
print("rccl 1-rank communicator ok", g.Objective())
g.FinalizeOptimization(); g.close()
dist.destroy_process_group()
