"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
import this module.  It loads oracle/liboracle.so (own restatement, built by
oracle/Makefile) and, when present, oracle/_ref/libref_tree.so (the reference's
own Tree.cpp compiled from /root/reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def _load():
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.oracle_lcg_fill.restype = C.c_uint
    lib.oracle_lcg_fill.argtypes = [C.c_uint, C.c_longlong, _fp]
    lib.oracle_grid_edges.restype = C.c_int
    lib.oracle_grid_edges.argtypes = [C.c_int, C.c_int, _ip]
    lib.oracle_edge_create.restype = C.c_void_p
    lib.oracle_edge_create.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _fp, _fp,
                                       C.c_float, C.c_float, C.c_float]
    for n in ("destroy", "init", "compute", "update"):
        f = getattr(lib, "oracle_edge_" + n)
        f.restype = None
        f.argtypes = [C.c_void_p]
    lib.oracle_edge_objective.restype = C.c_double
    lib.oracle_edge_objective.argtypes = [C.c_void_p]
    lib.oracle_edge_solve_one.restype = None
    lib.oracle_edge_solve_one.argtypes = [C.c_int, _fp, _fp, _fp, _fp, _fp, _fp,
                                          C.c_float, C.c_float, C.c_float, _fp, _fp]
    lib.oracle_edge_vertcnt.restype = _ip
    lib.oracle_edge_vertcnt.argtypes = [C.c_void_p]
    for n in ("mu_node", "pot_node", "m_node", "lambda_node", "mu_edge", "pot_edge", "vertmu"):
        f = getattr(lib, "oracle_edge_" + n)
        f.restype = _fp
        f.argtypes = [C.c_void_p]
    lib.oracle_edge_ranks_create.restype = C.c_void_p
    lib.oracle_edge_ranks_create.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _fp, _fp,
                                             C.c_float, C.c_float, C.c_float, C.c_int]
    for n in ("destroy", "init", "compute", "update"):
        f = getattr(lib, "oracle_edge_ranks_" + n)
        f.restype = None
        f.argtypes = [C.c_void_p]
    lib.oracle_edge_ranks_objective.restype = C.c_double
    lib.oracle_edge_ranks_objective.argtypes = [C.c_void_p]
    lib.oracle_edge_ranks_state.restype = C.c_void_p
    lib.oracle_edge_ranks_state.argtypes = [C.c_void_p]
    lib.oracle_edge_ranks_nverts.restype = C.c_int
    lib.oracle_edge_ranks_nverts.argtypes = [C.c_void_p, C.c_int]
    for n in ("verts", "vertcnt"):
        f = getattr(lib, "oracle_edge_ranks_" + n)
        f.restype = _ip
        f.argtypes = [C.c_void_p, C.c_int]
    lib.oracle_edge_ranks_vertmu.restype = _fp
    lib.oracle_edge_ranks_vertmu.argtypes = [C.c_void_p, C.c_int]
    lib.oracle_set_threads.restype = None
    lib.oracle_set_threads.argtypes = [C.c_int]
    lib.oracle_chain_create.restype = C.c_void_p
    lib.oracle_chain_create.argtypes = [C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_float]
    for n in ("destroy", "init", "compute", "update"):
        f = getattr(lib, "oracle_chain_" + n)
        f.restype = None
        f.argtypes = [C.c_void_p]
    for n in ("compute_range", "update_range"):
        f = getattr(lib, "oracle_chain_" + n)
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.oracle_chain_objective.restype = None
    lib.oracle_chain_objective.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.oracle_chain_minsum_batch.restype = None
    lib.oracle_chain_minsum_batch.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, _ip, _fp]
    lib.oracle_chain_labels.restype = None
    lib.oracle_chain_labels.argtypes = [C.c_void_p, _ip]
    for n in ("xrow", "xcol"):
        f = getattr(lib, "oracle_chain_" + n)
        f.restype = _ip
        f.argtypes = [C.c_void_p]
    for n in ("consensus", "lambda_row", "lambda_col", "theta"):
        f = getattr(lib, "oracle_chain_" + n)
        f.restype = _fp
        f.argtypes = [C.c_void_p]
    return lib


lib = _load()


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


# ---------------------------------------------------------------- inputs

def set_threads(n):
    lib.oracle_set_threads(int(n))


def lcg_fill(seed, n):
    """n uniform [0,1) floats from the 32-bit LCG; returns (array, new_state)."""
    out = np.empty(int(n), dtype=np.float32)
    st = lib.oracle_lcg_fill(C.c_uint(seed & 0xFFFFFFFF), int(n), _f(out))
    return out, st


def grid_edges(R, Cc):
    E = R * (Cc - 1) + Cc * (R - 1)
    el = np.empty(2 * max(E, 1), dtype=np.int32)
    n = lib.oracle_grid_edges(R, Cc, _i(el))
    assert n == E
    return el[: 2 * E].copy()


def pairwise_table(kind, k, seed_state=None):
    """Pairwise cost tables of the benchmark configurations."""
    a = np.arange(k, dtype=np.float32)[:, None]
    b = np.arange(k, dtype=np.float32)[None, :]
    if kind == "potts":
        return (0.5 * (a != b)).astype(np.float32)
    if kind == "trunc_linear":
        return (np.minimum(np.abs(a - b), 4.0) * 0.25).astype(np.float32)
    if kind == "dense":
        w, _ = lcg_fill(seed_state, k * k)
        return w.reshape(k, k).copy()
    raise ValueError(kind)


def synth_problem(R, Cc, k, seed=1234, pairwise="potts"):
    """Unary costs [R*C, k] and the k x k pairwise table."""
    un, st = lcg_fill(seed, R * Cc * k)
    W = pairwise_table(pairwise, k, st)
    return un.reshape(R * Cc, k), W


# ---------------------------------------------------------------- edge solver

class EdgeOracle:
    """Reference algorithm (one tree per edge), own restatement."""

    def __init__(self, V, edge_list, k, vert_pot, edge_pot, penalty=0.5, penalty_prime=1.0, rho=1.0):
        self.V, self.k = int(V), int(k)
        self.edge_list = np.ascontiguousarray(edge_list, dtype=np.int32)
        self.E = self.edge_list.size // 2
        vp = np.ascontiguousarray(vert_pot, dtype=np.float32).reshape(-1)
        ep = np.ascontiguousarray(edge_pot, dtype=np.float32).reshape(-1)
        assert vp.size == self.V * k and ep.size == self.E * k * k
        self.h = lib.oracle_edge_create(self.V, self.E, k, _i(self.edge_list), _f(vp), _f(ep),
                                        penalty, penalty_prime, rho)

    def __del__(self):
        if getattr(self, "h", None):
            lib.oracle_edge_destroy(self.h)
            self.h = None

    def init(self):
        lib.oracle_edge_init(self.h)

    def compute(self):
        lib.oracle_edge_compute(self.h)

    def update(self):
        lib.oracle_edge_update(self.h)

    def objective(self):
        return lib.oracle_edge_objective(self.h)

    def _view(self, name, n):
        p = getattr(lib, "oracle_edge_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=(n,))

    @property
    def vertcnt(self):
        return np.ctypeslib.as_array(lib.oracle_edge_vertcnt(self.h), shape=(self.V,))

    def state(self):
        E, k = self.E, self.k
        return {
            "mu_node": self._view("mu_node", E * 2 * k).copy(),
            "pot_node": self._view("pot_node", E * 2 * k).copy(),
            "m_node": self._view("m_node", E * 2 * k).copy(),
            "lambda_node": self._view("lambda_node", E * 2 * k).copy(),
            "mu_edge": self._view("mu_edge", E * k * k).copy(),
            "pot_edge": self._view("pot_edge", E * k * k).copy(),
            "vertmu": self._view("vertmu", self.V * k).copy(),
        }


class EdgeRanksOracle:
    """The reference algorithm as it runs on P MPI ranks (EDGE partition), all
    ranks simulated in one process."""

    def __init__(self, V, edge_list, k, vert_pot, edge_pot, nranks, penalty=0.5, penalty_prime=1.0, rho=1.0):
        self.V, self.k, self.P = int(V), int(k), int(nranks)
        self.edge_list = np.ascontiguousarray(edge_list, dtype=np.int32)
        self.E = self.edge_list.size // 2
        vp = np.ascontiguousarray(vert_pot, dtype=np.float32).reshape(-1)
        ep = np.ascontiguousarray(edge_pot, dtype=np.float32).reshape(-1)
        assert vp.size == self.V * k and ep.size == self.E * k * k
        self.h = lib.oracle_edge_ranks_create(self.V, self.E, k, _i(self.edge_list), _f(vp), _f(ep),
                                              penalty, penalty_prime, rho, self.P)

    def __del__(self):
        if getattr(self, "h", None):
            lib.oracle_edge_ranks_destroy(self.h)
            self.h = None

    def init(self):
        lib.oracle_edge_ranks_init(self.h)

    def compute(self):
        lib.oracle_edge_ranks_compute(self.h)

    def update(self):
        lib.oracle_edge_ranks_update(self.h)

    def objective(self):
        return lib.oracle_edge_ranks_objective(self.h)

    def state(self):
        """Per-edge arrays, in global edge order."""
        g = lib.oracle_edge_ranks_state(self.h)
        E, k = self.E, self.k
        view = lambda name, n: np.ctypeslib.as_array(getattr(lib, "oracle_edge_" + name)(g), shape=(n,)).copy()
        return {"mu_node": view("mu_node", E * 2 * k), "pot_node": view("pot_node", E * 2 * k),
                "m_node": view("m_node", E * 2 * k), "lambda_node": view("lambda_node", E * 2 * k),
                "mu_edge": view("mu_edge", E * k * k), "pot_edge": view("pot_edge", E * k * k)}

    def rank_vertices(self, r):
        """(sorted global ids, global copy counts, consensus [n, k]) of rank r."""
        n = lib.oracle_edge_ranks_nverts(self.h, r)
        if n == 0:
            return np.empty(0, np.int32), np.empty(0, np.int32), np.empty((0, self.k), np.float32)
        return (np.ctypeslib.as_array(lib.oracle_edge_ranks_verts(self.h, r), shape=(n,)).copy(),
                np.ctypeslib.as_array(lib.oracle_edge_ranks_vertcnt(self.h, r), shape=(n,)).copy(),
                np.ctypeslib.as_array(lib.oracle_edge_ranks_vertmu(self.h, r), shape=(n * self.k,)).reshape(n, self.k).copy())


def edge_solve_one(k, mu, pot, m, lam, mue, pote, penalty=0.5, penalty_prime=1.0, rho=1.0, fn=None):
    mu = np.array(mu, dtype=np.float32)
    lam = np.array(lam, dtype=np.float32)
    mue = np.array(mue, dtype=np.float32)
    pot = np.ascontiguousarray(pot, dtype=np.float32)
    m = np.ascontiguousarray(m, dtype=np.float32)
    pote = np.ascontiguousarray(pote, dtype=np.float32)
    al = np.zeros(k, dtype=np.float32)
    ar = np.zeros(k, dtype=np.float32)
    (fn or lib.oracle_edge_solve_one)(k, _f(mu), _f(pot), _f(m), _f(lam), _f(mue), _f(pote),
                                      penalty, penalty_prime, rho, _f(al), _f(ar))
    return mu, lam, mue, al, ar


# ---------------------------------------------------------------- chain solver

class ChainOracle:
    """Row/column chain min-sum ADMM (no reference counterpart)."""

    def __init__(self, R, Cc, k, unary, W, penalty=0.5, rho=1.0):
        self.R, self.C, self.k = int(R), int(Cc), int(k)
        un = np.ascontiguousarray(unary, dtype=np.float32).reshape(-1)
        w = np.ascontiguousarray(W, dtype=np.float32).reshape(-1)
        assert un.size == R * Cc * k and w.size == k * k
        self.h = lib.oracle_chain_create(R, Cc, k, _f(un), _f(w), penalty, rho)

    def __del__(self):
        if getattr(self, "h", None):
            lib.oracle_chain_destroy(self.h)
            self.h = None

    def init(self):
        lib.oracle_chain_init(self.h)

    def compute(self, r0=None, r1=None, c0=None, c1=None):
        if r0 is None:
            lib.oracle_chain_compute(self.h)
        else:
            lib.oracle_chain_compute_range(self.h, r0, r1, c0, c1)

    def update(self, r0=None, r1=None, c0=None, c1=None):
        if r0 is None:
            lib.oracle_chain_update(self.h)
        else:
            lib.oracle_chain_update_range(self.h, r0, r1, c0, c1)

    def objective(self, dual=True):
        p, d, l = C.c_double(), C.c_double(), C.c_double()
        lib.oracle_chain_objective(self.h, C.byref(p), C.byref(d) if dual else None, C.byref(l))
        return p.value, (d.value if dual else float("nan")), l.value

    def labels(self):
        out = np.empty(self.R * self.C, dtype=np.int32)
        lib.oracle_chain_labels(self.h, _i(out))
        return out.reshape(self.R, self.C)

    def _fview(self, name):
        p = getattr(lib, "oracle_chain_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=(self.R, self.C, self.k))

    def _iview(self, name):
        p = getattr(lib, "oracle_chain_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=(self.R, self.C))

    xrow = property(lambda s: s._iview("xrow"))
    xcol = property(lambda s: s._iview("xcol"))
    consensus = property(lambda s: s._fview("consensus"))
    lambda_row = property(lambda s: s._fview("lambda_row"))
    lambda_col = property(lambda s: s._fview("lambda_col"))
    theta = property(lambda s: s._fview("theta"))


def chain_minsum_batch(th, W):
    """The chain subproblem alone: th [n, T, k] adjusted unaries, W [k, k].  Returns
    (labels [n, T] int32, chain minima [n] float32) of the two-ended sweep."""
    th = np.ascontiguousarray(th, dtype=np.float32)
    n, T, k = th.shape
    w = np.ascontiguousarray(W, dtype=np.float32).reshape(-1)
    assert w.size == k * k
    x = np.empty((n, T), dtype=np.int32)
    val = np.empty(n, dtype=np.float32)
    lib.oracle_chain_minsum_batch(_f(th), _f(w), k, T, n, _i(x), _f(val))
    return x, val


# ---------------------------------------------------------------- reference Tree.cpp

def load_ref():
    """oracle/_ref/libref_tree.so (reference Tree.cpp) or None when not built."""
    path = os.path.join(_HERE, "_ref", "libref_tree.so")
    if not os.path.exists(path):
        return None
    r = C.CDLL(path)
    r.ref_create.restype = C.c_void_p
    r.ref_create.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _fp, _fp, C.c_float, C.c_float, C.c_float]
    for n in ("destroy", "init", "compute", "update"):
        f = getattr(r, "ref_" + n)
        f.restype = None
        f.argtypes = [C.c_void_p]
    r.ref_objective.restype = C.c_double
    r.ref_objective.argtypes = [C.c_void_p]
    r.ref_get.restype = None
    r.ref_get.argtypes = [C.c_void_p, C.c_int, _fp]
    r.ref_get_vertcnt.restype = None
    r.ref_get_vertcnt.argtypes = [C.c_void_p, _ip]
    r.ref_solve_one.restype = None
    r.ref_solve_one.argtypes = [C.c_int, _fp, _fp, _fp, _fp, _fp, _fp,
                                C.c_float, C.c_float, C.c_float, _fp, _fp]
    return r


class RefEdge:
    """The reference's Edge class driven through oracle/_ref."""
    NAMES = ["mu_node", "pot_node", "m_node", "lambda_node", "mu_edge", "pot_edge", "vertmu"]

    def __init__(self, r, V, edge_list, k, vert_pot, edge_pot, penalty=0.5, penalty_prime=1.0, rho=1.0):
        self.r, self.V, self.k = r, int(V), int(k)
        self.edge_list = np.ascontiguousarray(edge_list, dtype=np.int32)
        self.E = self.edge_list.size // 2
        vp = np.ascontiguousarray(vert_pot, dtype=np.float32).reshape(-1)
        ep = np.ascontiguousarray(edge_pot, dtype=np.float32).reshape(-1)
        self.h = r.ref_create(self.V, self.E, k, _i(self.edge_list), _f(vp), _f(ep),
                              penalty, penalty_prime, rho)

    def __del__(self):
        if getattr(self, "h", None):
            self.r.ref_destroy(self.h)
            self.h = None

    def init(self):
        self.r.ref_init(self.h)

    def compute(self):
        self.r.ref_compute(self.h)

    def update(self):
        self.r.ref_update(self.h)

    def objective(self):
        return self.r.ref_objective(self.h)

    def state(self):
        E, k, V = self.E, self.k, self.V
        sizes = [E * 2 * k] * 4 + [E * k * k] * 2 + [V * k]
        out = {}
        for i, (n, s) in enumerate(zip(self.NAMES, sizes)):
            a = np.empty(s, dtype=np.float32)
            self.r.ref_get(self.h, i, _f(a))
            out[n] = a
        return out

    @property
    def vertcnt(self):
        a = np.empty(self.V, dtype=np.int32)
        self.r.ref_get_vertcnt(self.h, _i(a))
        return a
