"""World-size-2 and -3 (gloo, CPU) test of the strip decomposition: every rank solves
its row and column chains with the oracle, the labels cross ranks through the
exchange plan of admm_amd.strips, and the result must equal the one-rank run
bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, R, C, k, iters, pairwise, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from admm_amd import strips as P
    un, W = O.synth_problem(R, C, k, 1234, pairwise)
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    st = P.strips(R, C, world)
    r0, nr, c0, nc = st[rank]
    for _ in range(iters):
        o.compute(r0, r0 + nr, c0, c0 + nc)
        a, b = P.pack(o.xrow[r0:r0 + nr, :], np.ascontiguousarray(o.xcol[:, c0:c0 + nc].T), R, C, world)
        send = [torch.from_numpy(np.concatenate([x.reshape(-1), y.reshape(-1)])) for x, y in zip(a, b)]
        recv = [torch.empty(st[h][1] * nc + st[h][3] * nr, dtype=torch.int32) for h in range(world)]
        # pairwise exchange (gloo has no list all_to_all on every build)
        reqs = []
        for h in range(world):
            if h == rank:
                recv[h].copy_(send[h])
            else:
                reqs.append(dist.isend(send[h], h))
                reqs.append(dist.irecv(recv[h], h))
        for q in reqs:
            q.wait()
        ra = [recv[h][: st[h][1] * nc].numpy() for h in range(world)]
        rb = [recv[h][st[h][1] * nc:].numpy() for h in range(world)]
        xrow_c, xcolT_r = P.unpack(ra, rb, R, C, world, rank)
        o.xrow[:, c0:c0 + nc] = xrow_c
        o.xcol[r0:r0 + nr, :] = xcolT_r.T
        o.update(r0, r0 + nr, 0, C)
        o.update(0, R, c0, c0 + nc)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), labels=o.labels()[r0:r0 + nr],
             lam_row=o.lambda_row[r0:r0 + nr], lam_col=o.lambda_col[:, c0:c0 + nc], r0=r0, c0=c0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,R,C,k,pairwise", [(2, 10, 12, 4, "potts"), (3, 11, 7, 8, "dense"), (2, 5, 9, 3, "dense")])
def test_strips_equal_single_rank(tmp_path, world, R, C, k, pairwise):
    from oracle import oracle as O
    iters = 8
    port = _free_port()
    mp.spawn(_worker, args=(world, port, R, C, k, iters, pairwise, str(tmp_path)), nprocs=world, join=True)
    un, W = O.synth_problem(R, C, k, 1234, pairwise)
    ref = O.ChainOracle(R, C, k, un, W)
    ref.init()
    for _ in range(iters):
        ref.compute()
        ref.update()
    for rank in range(world):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        r0, c0 = int(g["r0"]), int(g["c0"])
        nr, nc = g["labels"].shape[0], g["lam_col"].shape[1]
        assert np.array_equal(g["labels"], ref.labels()[r0:r0 + nr])
        assert np.array_equal(g["lam_row"], ref.lambda_row[r0:r0 + nr])
        assert np.array_equal(g["lam_col"], ref.lambda_col[:, c0:c0 + nc])


# ---- edge solver on several ranks: the exchange plan, host arithmetic only ----

def _plan_cases():
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    yield "grid 9x7", 63, O.grid_edges(9, 7)
    yield "grid 2x2", 4, O.grid_edges(2, 2)
    V = 40
    el = np.array([(i, i + 1) for i in range(V - 1)] + [(int(a), int(b)) for a, b in rng.integers(0, V, size=(30, 2)) if a != b],
                  dtype=np.int32).reshape(-1)
    yield "chain plus random chords", V, el
    yield "two components with a gap", 20, np.array([0, 1, 1, 2, 2, 3, 10, 11, 11, 12, 12, 19], dtype=np.int32)


@pytest.mark.parametrize("P", [1, 2, 3, 4, 7, 16])
def test_edge_exchange_plan_matches_the_restatement(P):
    """admm_edge_plan (what every rank builds in admm_create, no device needed) against the
    P-rank restatement of GetVerts / InitNeighborsByEdge (Grid.cpp:815-856, 198-331): edge
    slices, vertex sets, global copy counts; and the plans of two neighbouring ranks list
    the same shared vertices in the same order, which is what makes their messages match."""
    import admm_amd as A
    from oracle import oracle as O
    for name, V, el in _plan_cases():
        E = el.size // 2
        k = 2
        o = O.EdgeRanksOracle(V, el, k, np.zeros((V, k), np.float32), np.zeros((E, k * k), np.float32), P)
        plans = [A.edge_plan(el, P, r) for r in range(P)]
        seen = 0
        for r, pl in enumerate(plans):
            assert pl["edge0"] == seen, name
            seen += pl["nedges"]
            ids, cnt, _ = o.rank_vertices(r)
            assert np.array_equal(pl["verts"], ids), name
            assert np.array_equal(pl["counts"], cnt), name
            assert list(pl["neighbours"]) == sorted(pl["neighbours"]) and r not in pl["neighbours"], name
            for q, h in enumerate(pl["neighbours"]):
                other = plans[h]
                assert r in other["neighbours"], name
                qq = list(other["neighbours"]).index(r)
                mine = pl["verts"][pl["shared"][q]]
                theirs = other["verts"][other["shared"][qq]]
                assert np.array_equal(mine, theirs) and np.array_equal(mine, np.sort(mine)), name
                # what I am told the neighbour holds: the vertex's copies in ITS edge slice
                sl = el[2 * other["edge0"]:2 * (other["edge0"] + other["nedges"])]
                assert np.array_equal(pl["shared_counts"][q], np.array([(sl == v).sum() for v in mine], dtype=np.int32)), name
        assert seen == E, name
        # global copy count = degree in the whole graph
        deg = np.bincount(el, minlength=V)
        for pl in plans:
            assert np.array_equal(pl["counts"], deg[pl["verts"]]), name
