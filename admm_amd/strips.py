"""Strip partition and label-exchange plan of the chain solver (host logic).

Rank g of G owns the row chains of rows [r0_g, r0_g + nr_g) and the column
chains of columns [c0_g, c0_g + nc_g), both split as Grid::PartitionByRow
splits rows (Grid.cpp:377-390).  After Compute it holds
  xrow  [nr_g][cols]   labels of its row chains, and
  xcolT [nc_g][rows]   labels of its column chains (chain-major);
Communicate sends to every rank h
  a: xrow[:, c0_h : c0_h + nc_h]      (h needs them for its column chains)
  b: xcolT[:, r0_h : r0_h + nr_h]     (h needs them for its row chains)
and receives h's blocks, which land contiguously:
  a from h -> rows [r0_h, r0_h + nr_h) of xrow_on_my_cols  [rows][nc_g]
  b from h -> rows [c0_h, c0_h + nc_h) of xcolT_on_my_rows [cols][nr_g]
This module is the numpy statement of that plan; libadmm_hip.so executes the
same plan with RCCL send/recv (admm_communicate).
"""
import numpy as np

from .grid import partition


def strips(rows, cols, nranks):
    """[(r0, nr, c0, nc)] for every rank."""
    out = []
    for g in range(nranks):
        r0, nr = partition(rows, nranks, g)
        c0, nc = partition(cols, nranks, g)
        out.append((r0, nr, c0, nc))
    return out


def pack(xrow, xcolT, rows, cols, nranks):
    """Send blocks of one rank: ([a_h], [b_h]) for h in range(nranks)."""
    st = strips(rows, cols, nranks)
    a = [np.ascontiguousarray(xrow[:, c0:c0 + nc]) for (_, _, c0, nc) in st]
    b = [np.ascontiguousarray(xcolT[:, r0:r0 + nr]) for (r0, nr, _, _) in st]
    return a, b


def unpack(recv_a, recv_b, rows, cols, nranks, rank):
    """Received blocks -> (xrow_on_my_cols [rows][nc], xcolT_on_my_rows [cols][nr])."""
    st = strips(rows, cols, nranks)
    _, nr, _, nc = st[rank]
    xrow_c = np.empty((rows, nc), dtype=np.int32)
    xcolT_r = np.empty((cols, nr), dtype=np.int32)
    for h, (r0, nrh, c0, nch) in enumerate(st):
        xrow_c[r0:r0 + nrh, :] = np.asarray(recv_a[h]).reshape(nrh, nc)
        xcolT_r[c0:c0 + nch, :] = np.asarray(recv_b[h]).reshape(nch, nr)
    return xrow_c, xcolT_r
