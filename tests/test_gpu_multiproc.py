"""Two processes, one strip each, sharing the GPU: the strip path through the C ABI with
the label exchange staged through the host and moved by gloo.  (RCCL refuses two ranks on
one device, so on a one-GPU box this is the transport that can be exercised; the RCCL
transport runs the same pack / unpack plan.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, R, C, k, iters, pw, exchange, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import admm_amd as A  # binds its HIP runtime locally, so torch (gloo only here) can be loaded beside it
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    un, W = O.synth_problem(R, C, k, 1234, pw)
    g = A.Grid(rank=rank, nprocs=world, device=0)
    g.exchange = exchange
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k)
    g.Load(un, W)
    g.InitOptimization()
    lin = []
    for _ in range(iters):
        g.Compute()
        g.CommunicateHost(dist)
        g.Update()
        lin.append(g.ComputeLinearObj())
    p, d = g.Objective()
    r0, nr, c0, nc = g.StripInfo()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), labels=g.Labels().reshape(nr, C), lam=g.State(10),
             lin=np.array(lin), pd=np.array([p, d]), r0=r0)
    g.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,R,C,k,pw,exchange", [(2, 40, 70, 32, "potts", 0), (3, 33, 20, 8, "dense", 0),
                                                     (2, 21, 34, 16, "trunc_linear", 1)])
def test_two_processes_one_strip_each(tmp_path, world, R, C, k, pw, exchange):
    import multiprocessing as mp  # torch stays out of this process: the workers load it after the HIP library
    from oracle import oracle as O
    iters = 6
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, R, C, k, iters, pw, exchange, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0, "worker exit code %r" % (p.exitcode,)
    un, W = O.synth_problem(R, C, k, 1234, pw)
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    lin_ref = []
    for _ in range(iters):
        o.compute(); o.update()
        lin_ref.append(o.objective()[2])
    po, do, _ = o.objective()
    lin = np.zeros(iters)
    pd = np.zeros(2)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        r0, nr = int(z["r0"]), z["labels"].shape[0]
        assert np.array_equal(z["labels"], o.labels()[r0:r0 + nr])
        assert np.array_equal(z["lam"].reshape(nr, C, k), o.lambda_row[r0:r0 + nr])
        lin += z["lin"]
        pd += z["pd"]
    np.testing.assert_allclose(lin, np.array(lin_ref), rtol=1e-9)
    np.testing.assert_allclose(pd, np.array([po, do]), rtol=1e-9)


def _edge_worker(rank, world, port, R, C, k, iters, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import admm_amd as A
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    un, W = O.synth_problem(R, C, k, 1234, "potts")
    g = A.Grid(rank=rank, nprocs=world, device=0)
    g.SetPartitioner(A.EDGE)  # admm.cpp:41
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, C, k)
    g.Load(un, W)
    g.InitOptimization()
    objs = []
    for _ in range(iters):
        g.Compute()
        g.CommunicateHost(dist)
        g.Update()
        objs.append(g.ComputeLinearObj())
    np.savez(os.path.join(out_dir, "edge_rank%d.npz" % rank), obj=np.array(objs), m=g.Consensus(), ids=g.EdgeLocalVertices())
    g.close()
    dist.barrier()
    dist.destroy_process_group()


def test_edge_solver_one_process_per_rank(tmp_path):
    """The reference's own launch mode (mpiexec -n P, EDGE partition), with gloo carrying
    the messages of CommunicateByEdge (Grid.cpp:1639-1663)."""
    import multiprocessing as mp
    from oracle import oracle as O
    world, R, C, k, iters = 3, 10, 9, 4, 6
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_edge_worker, args=(r, world, port, R, C, k, iters, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0, "worker exit code %r" % (p.exitcode,)
    un, W = O.synth_problem(R, C, k, 1234, "potts")
    el = O.grid_edges(R, C)
    o = O.EdgeRanksOracle(R * C, el, k, un, np.tile(W.reshape(-1), el.size // 2), world)
    o.init()
    ref = []
    for _ in range(iters):
        o.compute(); o.update()
        ref.append(o.objective())
    obj = np.zeros(iters)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "edge_rank%d.npz" % rank))
        obj += z["obj"]
        ids, _, m = o.rank_vertices(rank)
        assert np.array_equal(z["ids"], ids)
        np.testing.assert_allclose(z["m"], m, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(obj, np.array(ref), rtol=1e-5)
