"""ctypes binding of libadmm_hip.so (declared in include/admm_hip.h).

There is no CPU fallback: if the library is missing, importing this module
raises, and every compute call needs a HIP device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadmm_hip.so")

SOLVER_EDGE_BETHE = 0
SOLVER_CHAIN_MINSUM = 1
EXCHANGE_LABELS = 0
EXCHANGE_DENSE = 1
COMM_ID_BYTES = 128
STATIC_LABELS_MAX = 4 << 20

# every symbol include/admm_hip.h declares
EXPORTS = [
    "admm_create", "admm_destroy", "admm_last_error", "admm_upload_unaries",
    "admm_upload_unaries_rows", "admm_upload_unaries_cols",
    "admm_upload_pairwise", "admm_init", "admm_compute", "admm_communicate",
    "admm_communicate_group", "admm_exchange_counts", "admm_exchange_pack_host",
    "admm_exchange_unpack_host", "admm_update",
    "admm_linear_objective", "admm_objective", "admm_objective_enqueue", "admm_objective_read", "admm_synchronize", "admm_query", "admm_finalize",
    "admm_download_labels", "admm_download_labels_static", "admm_download_consensus", "admm_download_state",
    "admm_state_size", "admm_partition", "admm_strip_info", "admm_edge_local_info", "admm_edge_local_vertices", "admm_edge_plan",
    "admm_comm_get_unique_id",
    "admm_comm_init", "admm_lcg_fill", "admm_grid_edges", "admm_profile_reset",
    "admm_profile_read", "admm_sweep_kernel_info", "admm_sweep_kernel_name", "admm_launch_info", "admm_device_count", "admm_device_memory", "admm_version",
    "admm_edge_log_values", "admm_edge_exp_values",
]


class AdmmDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_int32), ("cols", C.c_int32), ("k", C.c_int32), ("batch", C.c_int32),
        ("solver", C.c_int32), ("penalty", C.c_float), ("penalty_prime", C.c_float),
        ("rho", C.c_float), ("device", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32),
        ("exchange", C.c_int32), ("edge_list", C.POINTER(C.c_int32)),
        ("num_vertices", C.c_int32), ("num_edges", C.c_int32),
    ]


class AdmmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("admm_hip error %d: %s" % (code, msg))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C admm_amd/csrc` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # Default (local) binding on purpose: with RTLD_GLOBAL the HIP runtime and
    # RCCL this library brings in would also serve a later `import torch`,
    # which carries its own copies, and the mix aborts at exit.
    lib = C.CDLL(LIB_PATH)
    fp, ip, dp, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p
    sig = {
        "admm_create": (C.c_int, [C.POINTER(AdmmDesc), C.POINTER(vp)]),
        "admm_destroy": (C.c_int, [vp]),
        "admm_last_error": (C.c_char_p, [vp]),
        "admm_upload_unaries": (C.c_int, [vp, fp]),
        "admm_upload_unaries_rows": (C.c_int, [vp, C.c_int, C.c_int, fp]),
        "admm_upload_unaries_cols": (C.c_int, [vp, C.c_int, C.c_int, fp]),
        "admm_upload_pairwise": (C.c_int, [vp, fp, C.c_int]),
        "admm_init": (C.c_int, [vp]),
        "admm_compute": (C.c_int, [vp]),
        "admm_communicate": (C.c_int, [vp]),
        "admm_communicate_group": (C.c_int, [C.POINTER(vp), C.c_int]),
        "admm_exchange_counts": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "admm_exchange_pack_host": (C.c_int, [vp, ip]),
        "admm_exchange_unpack_host": (C.c_int, [vp, ip]),
        "admm_update": (C.c_int, [vp]),
        "admm_linear_objective": (C.c_int, [vp, dp]),
        "admm_objective": (C.c_int, [vp, dp, dp]),
        "admm_objective_enqueue": (C.c_int, [vp, C.c_int]),
        "admm_objective_read": (C.c_int, [vp, C.c_int, C.c_int, dp]),
        "admm_synchronize": (C.c_int, [vp]),
        "admm_query": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "admm_finalize": (C.c_int, [vp]),
        "admm_download_labels": (C.c_int, [vp, ip]),
        "admm_download_labels_static": (C.c_int, [vp, ip]),
        "admm_download_consensus": (C.c_int, [vp, fp]),
        "admm_download_state": (C.c_int, [vp, C.c_int, fp]),
        "admm_state_size": (C.c_size_t, [vp, C.c_int]),
        "admm_partition": (None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "admm_strip_info": (C.c_int, [vp] + [C.POINTER(C.c_int)] * 4),
        "admm_edge_local_info": (C.c_int, [vp] + [C.POINTER(C.c_int)] * 3),
        "admm_edge_local_vertices": (C.c_int, [vp, ip]),
        "admm_edge_plan": (C.c_int, [ip, C.c_int, C.c_int, C.c_int, ip, ip, ip, ip, ip, ip, ip]),
        "admm_comm_get_unique_id": (C.c_int, [vp]),
        "admm_comm_init": (C.c_int, [vp, vp]),
        "admm_lcg_fill": (C.c_uint32, [C.c_uint32, C.c_int64, fp]),
        "admm_grid_edges": (C.c_int, [C.c_int, C.c_int, ip]),
        "admm_profile_reset": (C.c_int, [vp, C.c_int]),
        "admm_profile_read": (C.c_int, [vp, dp, C.POINTER(C.c_int)]),
        "admm_sweep_kernel_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "admm_sweep_kernel_name": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "admm_launch_info": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "admm_device_count": (C.c_int, []),
        "admm_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "admm_edge_log_values": (C.c_int, [C.c_int, dp, dp, C.c_int64]),
        "admm_edge_exp_values": (C.c_int, [C.c_int, fp, dp, dp, C.POINTER(C.c_ubyte), C.c_int64]),
        "admm_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    return lib


lib = _load()
