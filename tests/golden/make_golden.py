#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ (run in the build container).

edge_*.npz come from the REFERENCE: the reference's own Tree.cpp compiled from
/root/reference by oracle/Makefile (oracle/_ref/libref_tree.so) and driven by
oracle/ref_tree_driver.cpp.  known_answers.json holds the totalObj values the
unmodified reference binary printed (64x64, k=4 and 3x3, k=2, LCG seed 1234,
Potts 0.5, penalties 0.5 / 1.0), which pin the loop around Tree.cpp as well.

chain_*.npz come from oracle/admm_oracle.c: the reference has no chain solver,
so these vectors only guard the restatement against accidental change.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def edge_case(name, R, C, k, iters, pairwise="potts", seed=1234, dump_at=()):
    ref = O.load_ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_tree.so is missing: run `make -C oracle` with /root/reference mounted")
    un, W = O.synth_problem(R, C, k, seed, pairwise)
    el = O.grid_edges(R, C)
    E = el.size // 2
    ep = np.tile(W.reshape(-1), E)
    r = O.RefEdge(ref, R * C, el, k, un, ep)
    r.init()
    out = {"R": R, "C": C, "k": k, "seed": seed, "vertcnt": r.vertcnt.copy(),
           "pot_node": r.state()["pot_node"], "pot_edge": r.state()["pot_edge"]}
    objs = []
    for it in range(1, iters + 1):
        r.compute()
        r.update()
        objs.append(r.objective())
        if it in dump_at:
            st = r.state()
            for n in ("mu_node", "lambda_node", "mu_edge", "m_node", "vertmu"):
                out["it%d_%s" % (it, n)] = st[n]
    out["objective"] = np.array(objs, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name), **out)
    return objs


def edge_graph_case(name):
    """A small non-grid graph (the reference accepts any edge list)."""
    ref = O.load_ref()
    V, k = 7, 3
    el = np.array([0, 1, 1, 2, 2, 0, 2, 3, 3, 4, 4, 5, 5, 3, 5, 6], dtype=np.int32)
    E = el.size // 2
    un, st = O.lcg_fill(77, V * k)
    ep, _ = O.lcg_fill(st, E * k * k)
    r = O.RefEdge(ref, V, el, k, un, ep)
    r.init()
    out = {"V": V, "k": k, "edge_list": el, "unary": un, "edge_pot": ep, "vertcnt": r.vertcnt.copy()}
    objs = []
    for it in range(1, 31):
        r.compute()
        r.update()
        objs.append(r.objective())
    st = r.state()
    for n in ("mu_node", "lambda_node", "mu_edge", "vertmu"):
        out["final_" + n] = st[n]
    out["objective"] = np.array(objs)
    np.savez_compressed(os.path.join(HERE, name), **out)


def chain_case(name, R, C, k, iters, pairwise, seed=1234):
    un, W = O.synth_problem(R, C, k, seed, pairwise)
    o = O.ChainOracle(R, C, k, un, W)
    o.init()
    out = {"R": R, "C": C, "k": k, "seed": seed, "pairwise": pairwise}
    prim, dual, lin = [], [], []
    for it in range(iters):
        o.compute()
        o.update()
        p, d, l = o.objective()
        prim.append(p); dual.append(d); lin.append(l)
    out.update(primal=np.array(prim), dual=np.array(dual), linear=np.array(lin),
               xrow=o.xrow.astype(np.int8), xcol=o.xcol.astype(np.int8), labels=o.labels().astype(np.int8),
               lambda_row=o.lambda_row.copy())
    np.savez_compressed(os.path.join(HERE, name), **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["chain"]:  # only the vectors of this repository's own chain restatement
        chain_case("chain_12x17_k4_potts.npz", 12, 17, 4, 30, "potts")
        chain_case("chain_20x9_k16_trunc.npz", 20, 9, 16, 20, "trunc_linear")
        chain_case("chain_10x70_k32_dense.npz", 10, 70, 32, 15, "dense")
        chain_case("chain_7x5_k5_dense.npz", 7, 5, 5, 20, "dense")
        chain_case("chain_2x1_k3_dense.npz", 2, 1, 3, 8, "dense")
        raise SystemExit(0)
    objs = edge_case("edge_3x3_k2.npz", 3, 3, 2, 3, dump_at=(1, 2, 3))
    print("3x3 k=2:", ["%.9f" % x for x in objs])
    objs = edge_case("edge_8x8_k4.npz", 8, 8, 4, 40, dump_at=(40,))
    objs64 = edge_case("edge_64x64_k4_obj.npz", 64, 64, 4, 500)
    print("64x64 k=4:", "%.6f %.6f %.6f" % (objs64[0], objs64[99], objs64[499]))
    edge_case("edge_6x5_k8_trunc.npz", 6, 5, 8, 25, pairwise="trunc_linear", dump_at=(25,))
    edge_graph_case("edge_graph7_k3.npz")
    chain_case("chain_12x17_k4_potts.npz", 12, 17, 4, 30, "potts")
    chain_case("chain_20x9_k16_trunc.npz", 20, 9, 16, 20, "trunc_linear")
    chain_case("chain_10x70_k32_dense.npz", 10, 70, 32, 15, "dense")
    chain_case("chain_7x5_k5_dense.npz", 7, 5, 5, 20, "dense")
    chain_case("chain_2x1_k3_dense.npz", 2, 1, 3, 8, "dense")  # chains of one and two nodes
    known = {
        "source": "totalObj lines printed by the unmodified reference binary (admm.cpp:97-99), "
                  "synthetic grid, LCG seed 1234, Potts 0.5, penalties (0.5, 1.0), 1 rank",
        "64x64_k4": {"1": 4570.997053, "2": 4020.227396, "3": 3465.771942, "100": 2024.650423,
                     "250": 2021.145384, "498": 2015.877310, "499": 2015.832952, "500": 2015.785300},
        "3x3_k2": {"1": 6.336054724, "2": 5.531242643, "3": 4.903375030,
                   "vertCnt": [2, 3, 2, 3, 4, 3, 2, 3, 2],
                   "edge0_x_potential_node": [0.357153833, 0.103506535, 0.249847278, 0.115354143],
                   "it3_m_v4": [0.508847475, 0.491152525],
                   "it3_edge0_mu": [0.279582053, 0.720417976, 0.310885429, 0.689114511],
                   "it3_edge0_lambda": [-0.00725613534, 0.00725615025, -0.0130264163, 0.0130263269]},
        "64x64_k4_2ranks": {"100": 2024.650422, "500": 2015.785286},
        "64x64_k4_4ranks": {"100": 2024.650435, "500": 2015.785304},
        "source_ranks": "the same run under mpiexec -n 2 and -n 4 (EDGE partition), values printed by the "
                        "unmodified reference binary",
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)
    print("golden vectors written to", HERE)
