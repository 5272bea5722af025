#!/usr/bin/env python3
"""Writes tests/golden/edge_log.npz: arguments and the bits of edge_log (admm_amd/csrc/edge_log.h)
as its host build gives them (tools/check_edge_log.cpp --values).  The GPU test asks the device
for the same arguments and expects the same bits; the CPU test rebuilds the host program and
expects this file."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def arguments():
    rng = np.random.default_rng(20240607)
    parts = [
        rng.random(8000, dtype=np.float32).astype(np.float64),                      # marginals
        (1.0 + (rng.random(4000) - 0.5) * 2.0 ** -5).astype(np.float32).astype(np.float64),  # near 1
        (1.0 + (rng.random(4000) - 0.5) * 2.0 ** -6),                               # near 1, binary64
        np.exp(rng.uniform(-700.0, 700.0, 4000)),                                   # any magnitude
        rng.integers(1, 0x7F800000, 4000, dtype=np.int64).astype(np.uint32).view(np.float32).astype(np.float64),
        np.array([1.0, np.float32(1e-17), 0.5, 2.0, 2.0 ** -149, 0.6875, 1.375, np.nextafter(1.0, 0.0), np.nextafter(1.0, 2.0),
                  1.0 - 2.0 ** -8, 1.0 + 2.0 ** -7, np.finfo(np.float64).tiny, np.finfo(np.float64).max, np.inf]),
    ]
    x = np.concatenate(parts)
    return np.where(x > 0, x, 1e-17)


def host_values(x, exe=None):
    with tempfile.TemporaryDirectory() as tmp:
        if exe is None:
            exe = os.path.join(tmp, "check_edge_log")
            subprocess.check_call(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-o", exe,
                                   os.path.join(ROOT, "tools", "check_edge_log.cpp")])
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        np.ascontiguousarray(x, dtype=np.float64).tofile(fin)
        subprocess.check_call([exe, "--values", fin, fout])
        return np.fromfile(fout, dtype=np.float64)


if __name__ == "__main__":
    x = arguments()
    y = host_values(x)
    np.savez_compressed(os.path.join(HERE, "edge_log.npz"), x=x, y_bits=y.view(np.uint64))
    err = np.abs(y - np.log(x)) / np.spacing(np.abs(np.log(x)))
    print("%d arguments; against numpy's log: worst %.2f ulp" % (x.size, np.nanmax(err[np.isfinite(err)])))
    sys.exit(0)
