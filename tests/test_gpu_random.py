"""Randomised parity: random shapes, label counts, tables (some quantised so that exact ties
abound), strip counts and exchange modes; labels of every iteration and the multipliers are
bit-compared with the oracle (tools/stress_parity.py runs longer sweeps of the same)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_cases_bit_exact(seed):
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(ROOT, "tools", "stress_parity.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(seed)
    for i in range(25):
        m.one(rng, i)
