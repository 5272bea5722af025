#!/usr/bin/env python3
"""Generates tests/golden/bench_objectives.json (run in the build container).

The trajectories bench.py times, solved by oracle/admm_oracle.c on the host: for every
chain workload of bench.py (LCG seed 1234, penalties 0.5 / 1.0 unless the name says
otherwise) the linear objective after every iteration and, at the iteration counts the
benchmark is run with (warmup + steps = 25 and 55; the headline workload also 100, 250 and
500, the length of the reference's own loop, admm.cpp:78), the decoded energy, the dual bound
and SHA-256 digests of the decoded labels, of the row chains' and the column chains'
labels (int32, little endian, grid-major) and of the row chains' multipliers (binary32).  bench.py reads
this file and compares what the HIP path left behind after its timed region with the
entry of its own iteration count; nothing of the oracle runs there.

The reference has no chain solver, so these values pin the HIP path to the CPU
restatement, not to the reference.

    python tests/golden/make_bench_objectives.py [workload ...]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(HERE, "bench_objectives.json")
CHECKPOINTS = (25, 55)
LONG_CHECKPOINTS = {"2048x2048_k32_potts": (25, 55, 100, 250, 500)}
SEED = 1234

# name: rows, cols, k, batch, pairwise, penalty
CASES = {
    "2048x2048_k32_potts": (2048, 2048, 32, 1, "potts", 0.5),
    "2048x2048_k32_potts_penalty0.3": (2048, 2048, 32, 1, "potts", 0.3),
    "2048x2048_k32_dense": (2048, 2048, 32, 1, "dense", 0.5),
    "512x512_k16_trunc_linear": (512, 512, 16, 1, "trunc_linear", 0.5),
    "1024x1024_k64_dense": (1024, 1024, 64, 1, "dense", 0.5),
    "256x(256x256)_k8_potts": (256, 256, 8, 256, "potts", 0.5),
    "64x64_k4_potts": (64, 64, 4, 1, "potts", 0.5),
}


def run_case(rows, cols, k, batch, pairwise, penalty, CHECKPOINTS=CHECKPOINTS):
    last = max(CHECKPOINTS)
    _, st0 = O.lcg_fill(SEED, rows * cols * k)
    W = O.pairwise_table(pairwise, k, st0)  # the table follows grid 0's draws (Grid.Generate)
    linear = np.zeros(last, dtype=np.float64)
    sums = {c: np.zeros(2, dtype=np.float64) for c in CHECKPOINTS}
    sha = {c: {n: hashlib.sha256() for n in ("labels", "xrow", "xcol", "lambda_row")} for c in CHECKPOINTS}
    for g in range(batch):
        un, _ = O.lcg_fill(SEED + g, rows * cols * k)
        o = O.ChainOracle(rows, cols, k, un, W, penalty=penalty, rho=1.0)
        o.init()
        for it in range(1, last + 1):
            o.compute()
            o.update()
            if it in CHECKPOINTS:
                p, d, l = o.objective(dual=True)
                sums[it] += (p, d)
                sha[it]["labels"].update(np.ascontiguousarray(o.labels(), dtype="<i4").tobytes())
                sha[it]["xrow"].update(np.ascontiguousarray(o.xrow, dtype="<i4").tobytes())
                sha[it]["xcol"].update(np.ascontiguousarray(o.xcol, dtype="<i4").tobytes())
                sha[it]["lambda_row"].update(np.ascontiguousarray(o.lambda_row, dtype="<f4").tobytes())
            else:
                _, _, l = o.objective(dual=False)
            linear[it - 1] += l
        del o
    return {
        "rows": rows, "cols": cols, "k": k, "batch": batch, "pairwise": pairwise,
        "penalty": penalty, "penalty_prime": 1.0, "rho": 1.0, "seed": SEED,
        "linear": [float(v) for v in linear],
        "at": {str(c): {"primal": float(sums[c][0]), "dual": float(sums[c][1]),
                        "linear": float(linear[c - 1]), "labels_sha256": sha[c]["labels"].hexdigest(),
                        "xrow_sha256": sha[c]["xrow"].hexdigest(), "xcol_sha256": sha[c]["xcol"].hexdigest(),
                        "lambda_row_sha256": sha[c]["lambda_row"].hexdigest()}
               for c in CHECKPOINTS},
    }


def main():
    O.set_threads(min(16, os.cpu_count() or 1))
    names = sys.argv[1:] or list(CASES)
    doc = {}
    if os.path.exists(OUT):
        doc = json.load(open(OUT))
    for n in names:
        t0 = time.time()
        doc[n] = run_case(*CASES[n], CHECKPOINTS=LONG_CHECKPOINTS.get(n, CHECKPOINTS))
        sys.stderr.write("%s: %.0f s\n" % (n, time.time() - t0))
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
