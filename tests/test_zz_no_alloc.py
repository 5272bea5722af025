"""admm_download_labels_static: the entry point that allocates no device memory, against
admm_download_labels (bit-exact, both are int32 labels), on two calls in a row, the second
on a smaller problem: the staging array is shared by every context on the GPU and keeps the
last call's contents."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _grid(R, Cc, k, seed, batch=1):
    import admm_amd as A
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, Cc, k, batch=batch)
    g.Generate(seed=seed, pairwise="potts")
    g.InitOptimization()
    for _ in range(3):
        g.Compute(); g.Communicate(); g.Update()
    return g


def test_static_staging_matches_context_staging_twice_second_smaller():
    big = _grid(300, 517, 16, 7)
    small = _grid(33, 70, 8, 11, batch=2)
    want_big, want_small = big.Labels(), small.Labels()
    assert want_big.max() > 0 and want_small.max() > 0  # not all zeros: stale contents could not pass for these
    got_big = big.LabelsStatic()
    got_small = small.LabelsStatic()   # smaller than the call before it: must not see its leftovers
    assert np.array_equal(got_big, want_big)
    assert np.array_equal(got_small, want_small)
    assert np.array_equal(big.LabelsStatic(), want_big)  # and back again
    big.close(); small.close()


def _free_bytes():
    from admm_amd._lib import lib
    f, t = C.c_uint64(), C.c_uint64()
    assert lib.admm_device_memory(0, C.byref(f), C.byref(t)) == 0
    return f.value


def test_static_call_allocates_no_device_memory():
    """Free device memory as the runtime reports it is the same before and after the static
    call on a context that has never staged anything; the allocating call on the same
    context then takes its staging (16 MiB for these 2048 x 2048 labels) and keeps it."""
    g = _grid(2048, 2048, 2, 3)
    g.Synchronize()
    before = _free_bytes()
    got = g.LabelsStatic()
    after = _free_bytes()
    assert after == before, (before, after)
    want = g.Labels()  # first staging of this context
    assert _free_bytes() <= before - (16 << 20) + (2 << 20), "the allocating call was expected to take 16 MiB of staging"
    assert np.array_equal(got, want)
    mid = _free_bytes()
    g.LabelsStatic()
    assert _free_bytes() == mid
    g.close()


def test_static_staging_refuses_what_does_not_fit():
    import admm_amd as A
    from admm_amd._lib import STATIC_LABELS_MAX, lib
    R, Cc = 2049, 2048  # one row more than the array holds
    assert R * Cc > STATIC_LABELS_MAX
    g = A.Grid()
    g.SetPartitioner(A.ROW)
    g.SetPenalties(0.5, 1.0)
    g.SetGlobalInfo(R, Cc, 2)
    g.Load(np.zeros((R * Cc, 2), dtype=np.float32), np.zeros((2, 2), dtype=np.float32))
    g.InitOptimization()
    out = np.empty(R * Cc, dtype=np.int32)
    rc = lib.admm_download_labels_static(g._h, out.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == -1 and b"static staging holds" in lib.admm_last_error(g._h)
    assert g.Labels().shape == (R, Cc)  # the allocating call still serves it
    g.close()
