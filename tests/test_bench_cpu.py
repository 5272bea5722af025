"""bench.py's host logic that needs no GPU: the check of a run against the committed oracle
trajectories (tests/golden/bench_objectives.json), the traffic model, the argument handling of
the self-launching multi-rank form."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _golden():
    return json.load(open(bench.GOLDEN))


def test_committed_trajectories_cover_every_chain_workload():
    doc = _golden()
    for name, (rows, cols, k, batch, pairwise) in bench.WORKLOADS.items():
        if name.startswith("edge_"):
            continue
        case = doc[name]
        assert (case["rows"], case["cols"], case["k"], case["batch"], case["pairwise"]) == (rows, cols, k, batch, pairwise)
        assert case["seed"] == 1234 and case["penalty"] == 0.5
        if name == "2048x2048_k32_potts":  # the headline also holds the reference's 500-iteration loop
            assert len(case["linear"]) == 500 and set(case["at"]) == {"25", "55", "100", "250", "500"}
        else:
            assert len(case["linear"]) == 55 and set(case["at"]) == {"25", "55"}
        for it, at in case["at"].items():
            assert at["linear"] == case["linear"][int(it) - 1]
            assert at["dual"] <= at["primal"]
            assert all(len(at[h]) == 64 for h in ("labels_sha256", "xrow_sha256", "xcol_sha256", "lambda_row_sha256"))
    assert "2048x2048_k32_potts_penalty0.3" in doc  # the binary32-multiplier run of the headline's `secondary`


def test_objective_check_accepts_the_committed_run_and_nothing_else():
    case = _golden()["2048x2048_k32_potts"]
    lin = np.array(case["linear"][5:25])
    at = case["at"]["25"]
    final = (at["primal"], at["linear"], at["dual"])
    ok = bench.objective_check("2048x2048_k32_potts", 0.5, 1234, 6, lin, 25, final, at["labels_sha256"])
    assert ok["checked"] and ok["ok"] and ok["labels_bit_exact"] and ok["rel_err"] == 0.0 and ok["iterations_compared"] == 20
    # a trace that is off by 1e-6 relative, a wrong labelling, a shifted iteration number: not ok
    assert not bench.objective_check("2048x2048_k32_potts", 0.5, 1234, 6, lin * (1 + 1e-6), 25, final, at["labels_sha256"])["ok"]
    assert not bench.objective_check("2048x2048_k32_potts", 0.5, 1234, 6, lin, 25, final, "0" * 64)["ok"]
    assert not bench.objective_check("2048x2048_k32_potts", 0.5, 1234, 7, lin, 26, None, None)["ok"]
    # no committed trajectory: says so instead of passing
    assert not bench.objective_check("2048x2048_k32_potts", 0.5, 99, 6, lin, 25, final, None)["checked"]
    assert not bench.objective_check("2048x2048_k32_potts", 0.25, 1234, 6, lin, 25, final, None)["checked"]
    # an iteration count without a checkpoint still compares the trace
    part = bench.objective_check("2048x2048_k32_potts", 0.5, 1234, 1, np.array(case["linear"][:10]), 10, final, None)
    assert part["checked"] and part["ok"] and "primal_at_10" not in part
    # penalty 0.3 has an entry of its own
    c3 = _golden()["2048x2048_k32_potts_penalty0.3"]
    assert bench.objective_check("2048x2048_k32_potts", 0.3, 1234, 1, np.array(c3["linear"][:12]))["ok"]


def test_traffic_model_of_the_headline_sweep():
    V = 2048 * 2048
    # int8 counts, two chains per wavefront, Potts masks + path-compression records, column pairs on one die
    assert abs(bench.stored_bytes(V, 32, 32, 2, True, 1, True, pcomp=True, pair_xcd=True) - 1.7029e9) < 1e6
    # ... and without that placement the column chains' 64-byte pieces cost whole lines
    assert abs(bench.stored_bytes(V, 32, 32, 2, True, 1, True, pcomp=True, pair_xcd=False) - 1.8371e9) < 1e6
    assert bench.sweep_bytes(V, 32) == V * (32 * 32 + 16) and bench.bytes_per_iteration(V, 32) == 4932501504


def test_requested_gpus_parsing():
    assert bench._requested_gpus(["--steps", "5"]) == 1
    assert bench._requested_gpus(["--gpus", "8", "--steps", "5"]) == 8
    assert bench._requested_gpus(["--steps=5", "--gpus=4"]) == 4
