"""ctypes binding of libadmm_graphfile.so (declared in include/admm_graphfile.h):
the reference's graph files (NetCDF classic family, schema of Grid::Save,
Grid.cpp:1066-1097) without PnetCDF.  Host only; needs no HIP device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadmm_graphfile.so")

# every symbol include/admm_graphfile.h declares
EXPORTS = [
    "admm_graphfile_read_info", "admm_graphfile_read_edges", "admm_graphfile_read_unaries",
    "admm_graphfile_read_pairwise", "admm_graphfile_read_unaries_block", "admm_graphfile_shared_table", "admm_graphfile_write",
    "admm_graphfile_convert", "admm_graphfile_last_error",
]


class GraphFileInfo(C.Structure):
    _fields_ = [
        ("version", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("num_vertices", C.c_int64), ("num_edges", C.c_int64), ("k", C.c_int32), ("k2", C.c_int32),
    ]

    def __repr__(self):
        return "GraphFileInfo(version=%d, rows=%d, cols=%d, num_vertices=%d, num_edges=%d, k=%d, k2=%d)" % (
            self.version, self.rows, self.cols, self.num_vertices, self.num_edges, self.k, self.k2)


class GraphFileError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("admm_graphfile error %d: %s" % (code, msg))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C admm_amd/csrc`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    fp, ip, cp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_char_p
    sig = {
        "admm_graphfile_read_info": (C.c_int, [cp, C.POINTER(GraphFileInfo)]),
        "admm_graphfile_read_edges": (C.c_int, [cp, C.c_int64, C.c_int64, ip]),
        "admm_graphfile_read_unaries": (C.c_int, [cp, C.c_int64, C.c_int64, fp, C.c_int]),
        "admm_graphfile_read_pairwise": (C.c_int, [cp, C.c_int64, C.c_int64, fp, C.c_int]),
        "admm_graphfile_read_unaries_block": (C.c_int, [cp] + [C.c_int32] * 5 + [fp, C.c_int]),
        "admm_graphfile_shared_table": (C.c_int, [cp, fp, C.c_int, C.POINTER(C.c_int)]),
        "admm_graphfile_write": (C.c_int, [cp, C.POINTER(GraphFileInfo), ip, fp, fp, C.c_int]),
        "admm_graphfile_convert": (C.c_int, [cp, cp]),
        "admm_graphfile_last_error": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    return lib


lib = _load()
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def _ck(rc):
    if rc != 0:
        raise GraphFileError(rc, (lib.admm_graphfile_last_error() or b"").decode("utf-8", "replace"))


def _path(p):
    return os.fsencode(os.fspath(p))


def read_info(path):
    """Grid::SetGlobalInfo(fname) (Grid.cpp:654-702)."""
    info = GraphFileInfo()
    _ck(lib.admm_graphfile_read_info(_path(path), C.byref(info)))
    return info


def read_edges(path, first=0, count=None):
    """Rows [first, first+count) of edgeList (Grid.cpp:476-494): int32 [count, 2]."""
    if count is None:
        count = read_info(path).num_edges - first
    out = np.empty((int(count), 2), dtype=np.int32)
    _ck(lib.admm_graphfile_read_edges(_path(path), int(first), int(count), out.ctypes.data_as(_ip)))
    return out


def read_unaries(path, first=0, count=None, negate=True):
    """Rows of verticePotentials (Grid.cpp:510-523); negate = the loader's sign flip (Grid.cpp:547-553)."""
    info = read_info(path)
    if count is None:
        count = info.num_vertices - first
    out = np.empty((int(count), info.k), dtype=np.float32)
    _ck(lib.admm_graphfile_read_unaries(_path(path), int(first), int(count), out.ctypes.data_as(_fp), 1 if negate else 0))
    return out


def read_unaries_block(path, grid_cols, row0, nrows, col0, ncols, negate=True):
    """A rectangle of a grid's verticePotentials, [nrows, ncols, k]: what one strip keeps."""
    info = read_info(path)
    out = np.empty((int(nrows), int(ncols), info.k), dtype=np.float32)
    _ck(lib.admm_graphfile_read_unaries_block(_path(path), int(grid_cols), int(row0), int(nrows), int(col0), int(ncols),
                                              out.ctypes.data_as(_fp), 1 if negate else 0))
    return out


def read_pairwise(path, first=0, count=None, negate=True):
    """Rows of edgePotentials (Grid.cpp:529-543); negate = the loader's sign flip (Grid.cpp:554-560)."""
    info = read_info(path)
    if count is None:
        count = info.num_edges - first
    out = np.empty((int(count), info.k2), dtype=np.float32)
    _ck(lib.admm_graphfile_read_pairwise(_path(path), int(first), int(count), out.ctypes.data_as(_fp), 1 if negate else 0))
    return out


def shared_table(path, negate=True):
    """(table of edge 0 as [k, k], whether every edge carries that same table)."""
    info = read_info(path)
    table = np.empty(info.k2, dtype=np.float32)
    shared = C.c_int()
    _ck(lib.admm_graphfile_shared_table(_path(path), table.ctypes.data_as(_fp), 1 if negate else 0, C.byref(shared)))
    return table.reshape(info.k, -1), bool(shared.value)


def write(path, num_vertices, edge_list, unary, pairwise, rows=-1, cols=-1, version=0):
    """Grid::Save (Grid.cpp:1049-1131).  Values are written as given (a file
    holds scores, the negated costs).  pairwise: [E, k*k], or one [k, k] table
    that is written for every edge."""
    edge_list = np.ascontiguousarray(edge_list, dtype=np.int32).reshape(-1, 2)
    unary = np.ascontiguousarray(unary, dtype=np.float32).reshape(int(num_vertices), -1)
    k = unary.shape[1] if num_vertices else int(round(np.sqrt(np.asarray(pairwise).size)))
    pairwise = np.ascontiguousarray(pairwise, dtype=np.float32).reshape(-1)
    E = edge_list.shape[0]
    per_edge = not (pairwise.size == k * k and E != 1)
    if per_edge and pairwise.size != E * k * k:
        raise ValueError("pairwise must hold k*k or E*k*k values")
    info = GraphFileInfo(int(version), int(rows), int(cols), int(num_vertices), E, k, k * k)
    _ck(lib.admm_graphfile_write(_path(path), C.byref(info), edge_list.ctypes.data_as(_ip),
                                 unary.ctypes.data_as(_fp), pairwise.ctypes.data_as(_fp), 1 if per_edge else 0))


def convert(manifest_path, out_path):
    """Grid::Convert + Save (Grid.cpp:985-1045, Converter.cpp:16-22)."""
    _ck(lib.admm_graphfile_convert(_path(manifest_path), _path(out_path)))
