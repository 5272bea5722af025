"""Host-side mirror of the reference's driver API (class Grid, Grid.h:98-137).

The method names and their order of use are the reference's (admm.cpp:41-101):
SetPartitioner, SetPenalties, SetGlobalInfo, Load / Generate, InitOptimization,
then per iteration Compute, Communicate, Update, ComputeLinearObj, and finally
FinalizeOptimization.  Every method forwards to the C ABI of libadmm_hip.so;
nothing is computed in Python.
"""
import ctypes as C
import os

import numpy as np

from . import _lib, graphfile
from ._lib import (AdmmDesc, AdmmError, EXCHANGE_DENSE, EXCHANGE_LABELS,
                   SOLVER_CHAIN_MINSUM, SOLVER_EDGE_BETHE, lib)

# enum Partitioner (Grid.h:21)
ROW, COL, BLOCK, EDGE = 0, 1, 2, 3

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def _f(a):
    return a.ctypes.data_as(_fp)


def partition(n, nranks, rank):
    """Grid::PartitionByRow's split (Grid.cpp:377-390): (start, count)."""
    s, c = C.c_int(), C.c_int()
    lib.admm_partition(n, nranks, rank, C.byref(s), C.byref(c))
    return s.value, c.value


def lcg_fill(state, n):
    out = np.empty(int(n), dtype=np.float32)
    st = lib.admm_lcg_fill(C.c_uint32(state & 0xFFFFFFFF), int(n), _f(out))
    return out, st


def grid_edges(rows, cols):
    E = rows * (cols - 1) + cols * (rows - 1)
    el = np.empty(2 * max(E, 1), dtype=np.int32)
    n = lib.admm_grid_edges(rows, cols, el.ctypes.data_as(_ip))
    assert n == E
    return el[:2 * E].copy()


def pairwise_table(kind, k, lcg_state=None):
    """The pairwise cost tables of the benchmark configurations."""
    a = np.arange(k, dtype=np.float32)[:, None]
    b = np.arange(k, dtype=np.float32)[None, :]
    if kind == "potts":
        return (0.5 * (a != b)).astype(np.float32)
    if kind == "trunc_linear":
        return (np.minimum(np.abs(a - b), 4.0) * 0.25).astype(np.float32)
    if kind == "dense":
        w, _ = lcg_fill(lcg_state, k * k)
        return w.reshape(k, k).copy()
    raise ValueError("unknown pairwise kind %r" % (kind,))


class Grid:
    """One strip of an ADMM MAP-inference problem on one GPU."""

    def __init__(self, rank=0, nprocs=1, device=0):
        self._rank, self._nprocs, self._device = int(rank), int(nprocs), int(device)
        self.part = EDGE
        self.penalty, self.penaltyPrime, self.rhoEdge = 0.1, 0.5, 1.0  # Grid.cpp:79-80
        self.rows = self.cols = self.k = 0
        self.batch = 1
        self.exchange = EXCHANGE_LABELS
        self._edge_list = None
        self._num_vertices = 0
        self._unary = None
        self._pairwise = None
        self._per_edge = False
        self._unary_file = None  # strips of a grid file: each strip reads its own share at InitOptimization
        self._h = None

    # ---- configuration (Grid.h:98-111) ----
    def GetRank(self):
        return self._rank

    def GetNprocs(self):
        return self._nprocs

    def SetPartitioner(self, part):
        """EDGE: the reference's solver (one tree per edge).  ROW: row and
        column chains over row / column strips."""
        if part not in (ROW, EDGE):
            raise ValueError("no implementation for this partitioner")  # Grid.cpp:1153-1158
        self.part = part

    def SetPenalties(self, p, pp):
        self.penalty, self.penaltyPrime = float(p), float(pp)

    def SetRhoEdge(self, rho):
        self.rhoEdge = float(rho)

    def SetGlobalInfo(self, rows, cols=None, vert_potential_size=None, edge_potential_size=None, batch=1):
        """SetGlobalInfo(rows, cols, k, k*k) (Grid.cpp:96-116), or
        SetGlobalInfo(fname): the sizes a graph file states (Grid.cpp:654-702)."""
        if isinstance(rows, (str, os.PathLike)):
            info = graphfile.read_info(rows)
            self.rows, self.cols, self.k, self.batch = info.rows, info.cols, info.k, 1
            self._file_info = info
            if info.k2 != info.k ** 2:
                raise ValueError("numEdgePotentials must be numVertPotentials squared")
            return
        if edge_potential_size is not None and edge_potential_size != vert_potential_size ** 2:
            raise ValueError("numEdgePotentials must be numVertPotentials squared")
        self.rows, self.cols, self.k, self.batch = int(rows), int(cols), int(vert_potential_size), int(batch)

    def SetGraph(self, num_vertices, edge_list):
        """Arbitrary edge list (what Grid::Load reads from a file, Grid.cpp:476-494)."""
        self._edge_list = np.ascontiguousarray(edge_list, dtype=np.int32).reshape(-1)
        self._num_vertices = int(num_vertices)

    # ---- data ----
    def Load(self, unary, pairwise=None, per_edge=False):
        """Load(fname): a graph file (Grid.cpp:445-565), sign-flipped into costs
        as the reference's loader does.  EDGE partitioner: any graph, one table
        per edge.  ROW: the file must hold a rows x cols grid in Generate's edge
        order whose edges all carry the same table.
        Load(unary, pairwise, per_edge): potentials as Grid holds them after
        the sign flip (Grid.cpp:547-560), i.e. costs.  unary [batch*rows*cols, k];
        pairwise either one k x k table or [E, k*k]."""
        if isinstance(unary, (str, os.PathLike)):
            return self._load_file(unary)
        self._unary_file = None
        self._unary = np.ascontiguousarray(unary, dtype=np.float32).reshape(-1)
        self._pairwise = np.ascontiguousarray(pairwise, dtype=np.float32).reshape(-1)
        self._per_edge = bool(per_edge)

    def _load_file(self, fname):
        info = graphfile.read_info(fname)
        if self.k == 0:
            self.SetGlobalInfo(fname)
        if (info.k, info.rows, info.cols) != (self.k, self.rows, self.cols):
            raise ValueError("SetGlobalInfo disagrees with %s" % (fname,))
        el = graphfile.read_edges(fname).reshape(-1)
        if self.part == EDGE:
            un = graphfile.read_unaries(fname, negate=True)
            self.SetGraph(info.num_vertices, el)
            self.Load(un, graphfile.read_pairwise(fname, negate=True), per_edge=True)
            return
        if info.rows < 1 or info.cols < 1 or info.rows * info.cols != info.num_vertices:
            raise ValueError("%s does not describe a rows x cols grid; use the EDGE partitioner" % (fname,))
        if not np.array_equal(el, grid_edges(info.rows, info.cols)):
            raise ValueError("the edge list of %s is not the grid's (Grid::Generate order); use the EDGE partitioner" % (fname,))
        table, shared = graphfile.shared_table(fname, negate=True)
        if not shared:
            raise ValueError("the edges of %s carry different pairwise tables; the chain solver needs one shared table" % (fname,))
        self._edge_list = None
        if self._nprocs > 1:
            # one strip of several: as a rank of the reference reads only its own
            # hyperslab (Grid.cpp:446-454, 510-523), this strip's rows and columns
            # are read, a bounded block at a time, when InitOptimization uploads them
            self._unary = None
            self._pairwise = np.ascontiguousarray(table, dtype=np.float32).reshape(-1)
            self._per_edge = False
            self._unary_file = os.fspath(fname)
            return
        self.Load(graphfile.read_unaries(fname, negate=True), table)

    def Save(self, fname, version=0):
        """Grid::Save (Grid.cpp:1049-1131): the potentials this object holds,
        sign-flipped back into the scores a graph file holds."""
        if self._unary is None and self._unary_file is None:
            raise AdmmError(-2, "Load or Generate before Save")
        if self.batch != 1:
            raise ValueError("a graph file holds one graph (batch == 1)")
        unary = self._unary
        if unary is None:
            # a strip of several that loaded from a file keeps no host copy of the unaries: read
            # them back, as every rank of the reference can save what it loaded
            unary = graphfile.read_unaries(self._unary_file, negate=True).reshape(-1)
        if self._edge_list is not None:
            el, V = self._edge_list, self._num_vertices
        else:
            el, V = grid_edges(self.rows, self.cols), self.rows * self.cols
        pw = -self._pairwise
        graphfile.write(fname, V, el, -unary, pw.reshape(-1, self.k * self.k) if self._per_edge else pw.reshape(self.k, self.k),
                        rows=self.rows, cols=self.cols, version=version)

    def Convert(self, manifest):
        """Grid::Convert (Grid.cpp:985-1045): raw binary files named by a four-line manifest
        -> this object; follow with Save, as Converter.cpp does.  The raw files hold what a
        graph file holds (scores), so this object keeps their negation, as after any Load."""
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, "converted.nc")
            graphfile.convert(manifest, tmp)
            keep = self.part
            self.part = EDGE
            try:
                self.k = 0
                self._load_file(tmp)
            finally:
                self.part = keep

    def Generate(self, seed=1234, pairwise="potts"):
        """Synthetic potentials: LCG unaries in [0,1) drawn node-major, grid g
        of a batch seeded with seed+g, and a named pairwise table."""
        n = self.rows * self.cols * self.k
        parts, st = [], None
        for g in range(self.batch):
            u, st_g = lcg_fill(seed + g, n)
            parts.append(u)
            if g == 0:
                st = st_g
        self.Load(np.concatenate(parts), pairwise_table(pairwise, self.k, st))

    # ---- optimisation (admm.cpp:73-101) ----
    UPLOAD_BLOCK_BYTES = 64 << 20  # host staging of a strip that reads its share from a file

    def _upload_shares_from_file(self):
        """This strip's rows, then its columns of every row, block by block."""
        _, nrows, c0, ncols = self.StripInfo()
        r0 = self.StripInfo()[0]
        k, cols = self.k, self.cols
        step = max(1, self.UPLOAD_BLOCK_BYTES // (4 * k * max(cols, 1)))
        for r in range(r0, r0 + nrows, step):
            n = min(step, r0 + nrows - r)
            blk = graphfile.read_unaries_block(self._unary_file, cols, r, n, 0, cols, negate=True)
            self._ck(lib.admm_upload_unaries_rows(self._h, r, n, _f(blk)))
        step = max(1, self.UPLOAD_BLOCK_BYTES // (4 * k * max(ncols, 1)))
        for r in range(0, self.rows, step):
            n = min(step, self.rows - r)
            blk = graphfile.read_unaries_block(self._unary_file, cols, r, n, c0, ncols, negate=True)
            self._ck(lib.admm_upload_unaries_cols(self._h, r, n, _f(blk)))

    def InitOptimization(self):
        if self._unary is None and self._unary_file is None:
            raise AdmmError(-2, "Load or Generate before InitOptimization")
        # sizes the C side will read: a mismatch would run past the numpy buffers
        graph = self._edge_list is not None
        if self.k < 1 or (not graph and (self.rows < 1 or self.cols < 1)):
            raise ValueError("SetGlobalInfo before InitOptimization")
        nv = self._num_vertices if graph else self.batch * self.rows * self.cols
        ne = self._edge_list.size // 2 if graph else self.rows * (self.cols - 1) + self.cols * (self.rows - 1)
        if self._unary is not None and self._unary.size != nv * self.k:
            raise ValueError("unary potentials hold %d values, %d vertices x %d labels expected" % (self._unary.size, nv, self.k))
        want_pw = ne * self.k * self.k if self._per_edge else self.k * self.k
        if self._pairwise is None or self._pairwise.size != want_pw:
            raise ValueError("pairwise potentials hold %d values, %d expected" % (0 if self._pairwise is None else self._pairwise.size, want_pw))
        self.close()  # a second InitOptimization replaces the context
        d = AdmmDesc()
        d.rows, d.cols, d.k, d.batch = self.rows, self.cols, self.k, self.batch
        d.solver = SOLVER_EDGE_BETHE if self.part == EDGE else SOLVER_CHAIN_MINSUM
        d.penalty, d.penalty_prime, d.rho = self.penalty, self.penaltyPrime, self.rhoEdge
        d.device, d.rank, d.nranks = self._device, self._rank, self._nprocs
        d.exchange = self.exchange
        if self._edge_list is not None:
            d.edge_list = self._edge_list.ctypes.data_as(_ip)
            d.num_vertices = self._num_vertices
            d.num_edges = self._edge_list.size // 2
        h = C.c_void_p()
        rc = lib.admm_create(C.byref(d), C.byref(h))
        if rc != 0:
            raise AdmmError(rc, (lib.admm_last_error(None) or b"").decode())
        self._h = h
        if self._unary is None:
            self._upload_shares_from_file()
        else:
            self._ck(lib.admm_upload_unaries(self._h, _f(self._unary)))
        self._ck(lib.admm_upload_pairwise(self._h, _f(self._pairwise), 1 if self._per_edge else 0))
        self._ck(lib.admm_init(self._h))

    def CommInit(self, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        self._ck(lib.admm_comm_init(self._h, buf))

    def Compute(self):
        self._ck(lib.admm_compute(self._h))

    def Communicate(self):
        self._ck(lib.admm_communicate(self._h))

    def CommunicateHost(self, dist):
        """Communicate with the labels staged through host memory and moved by
        torch.distributed point-to-point calls (any backend, e.g. gloo)."""
        n = self._nprocs
        if n == 1:
            return
        import torch
        sc = (C.c_int64 * n)()
        rcnt = (C.c_int64 * n)()
        self._ck(lib.admm_exchange_counts(self._h, sc, rcnt))
        send = np.empty(max(1, sum(sc)), dtype=np.int32)
        recv = np.empty(max(1, sum(rcnt)), dtype=np.int32)
        self._ck(lib.admm_exchange_pack_host(self._h, send.ctypes.data_as(_ip)))
        ts, tr = torch.from_numpy(send), torch.from_numpy(recv)
        so = np.concatenate([[0], np.cumsum(list(sc))]).astype(np.int64)
        ro = np.concatenate([[0], np.cumsum(list(rcnt))]).astype(np.int64)
        reqs = []
        for h in range(n):
            if h == self._rank:
                tr[ro[h]:ro[h + 1]] = ts[so[h]:so[h + 1]]
                continue
            if sc[h]:
                reqs.append(dist.isend(ts[so[h]:so[h + 1]], h))
            if rcnt[h]:
                reqs.append(dist.irecv(tr[ro[h]:ro[h + 1]], h))
        for q in reqs:
            q.wait()
        self._ck(lib.admm_exchange_unpack_host(self._h, recv.ctypes.data_as(_ip)))

    def Update(self):
        self._ck(lib.admm_update(self._h))

    def ComputeLinearObj(self):
        v = C.c_double()
        self._ck(lib.admm_linear_objective(self._h, C.byref(v)))
        return v.value

    def Objective(self, dual=True):
        p, d = C.c_double(), C.c_double(float("nan"))
        self._ck(lib.admm_objective(self._h, C.byref(p), C.byref(d) if dual else None))
        return p.value, d.value

    def ObjectiveEnqueue(self, slot):
        self._ck(lib.admm_objective_enqueue(self._h, int(slot)))

    def ObjectiveRead(self, slot, n):
        l = np.empty(n, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._ck(lib.admm_objective_read(self._h, int(slot), int(n), l.ctypes.data_as(dp)))
        return l

    def Synchronize(self):
        self._ck(lib.admm_synchronize(self._h))

    def Busy(self):
        """True while work queued on this strip's stream is still running (no waiting)."""
        b = C.c_int()
        self._ck(lib.admm_query(self._h, C.byref(b)))
        return bool(b.value)

    def SynchronizeWithin(self, seconds):
        """Synchronize() with a deadline: False if the stream is still busy after `seconds`."""
        import time
        t0 = time.monotonic()
        while self.Busy():
            if time.monotonic() - t0 > seconds:
                return False
            time.sleep(0.001)
        return True

    def FinalizeOptimization(self):
        if self._h:
            self._ck(lib.admm_finalize(self._h))

    # ---- results ----
    def StripInfo(self):
        v = [C.c_int() for _ in range(4)]
        self._ck(lib.admm_strip_info(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def EdgeLocalInfo(self):
        """Edge solver: (first edge, number of edges, number of distinct vertices) of this rank."""
        v = [C.c_int() for _ in range(3)]
        self._ck(lib.admm_edge_local_info(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def EdgeLocalVertices(self):
        """Edge solver: global ids of this rank's vertices, ascending (the rows of Consensus())."""
        n = self.EdgeLocalInfo()[2]
        out = np.empty(max(n, 1), dtype=np.int32)
        self._ck(lib.admm_edge_local_vertices(self._h, out.ctypes.data_as(_ip)))
        return out[:n]

    def Labels(self):
        _, nrows, _, _ = self.StripInfo()
        out = np.empty((self.batch, nrows, self.cols), dtype=np.int32)
        self._ck(lib.admm_download_labels(self._h, out.ctypes.data_as(_ip)))
        return out[0] if self.batch == 1 else out

    def LabelsStatic(self):
        """Labels() through the library's fixed staging array: no device allocation."""
        _, nrows, _, _ = self.StripInfo()
        out = np.empty((self.batch, nrows, self.cols), dtype=np.int32)
        self._ck(lib.admm_download_labels_static(self._h, out.ctypes.data_as(_ip)))
        return out[0] if self.batch == 1 else out

    def Consensus(self):
        if self.part == EDGE:
            V = self.EdgeLocalInfo()[2]
            out = np.empty((V, self.k), dtype=np.float32)
            if V == 0:
                return out
        else:
            _, nrows, _, _ = self.StripInfo()
            out = np.empty((self.batch * nrows * self.cols, self.k), dtype=np.float32)
        self._ck(lib.admm_download_consensus(self._h, _f(out)))
        return out

    def State(self, which):
        known = (0, 1, 2, 3, 4, 5) if self.part == EDGE else (10, 11, 12, 13, 14)
        if which not in known:
            raise AdmmError(-1, "unknown state id %d" % which)
        n = lib.admm_state_size(self._h, which)
        is_int = which in (13, 14)
        if n == 0:  # an empty strip
            return np.empty(0, dtype=np.int32 if is_int else np.float32)
        out = np.empty(n, dtype=np.int32 if is_int else np.float32)
        self._ck(lib.admm_download_state(self._h, which, C.cast(out.ctypes.data, _fp)))
        return out

    def ProfileReset(self, enable=True):
        """enable: False / True, or n > 1 to time every n-th Compute only."""
        self._ck(lib.admm_profile_reset(self._h, int(enable)))

    def ProfileRead(self):
        ms, n = C.c_double(), C.c_int()
        self._ck(lib.admm_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def SweepKernelInfo(self):
        """(lanes a label vector is padded to, lanes per target label, Potts kernel chosen)."""
        kp, ns, potts = C.c_int(), C.c_int(), C.c_int()
        self._ck(lib.admm_sweep_kernel_info(self._h, C.byref(kp), C.byref(ns), C.byref(potts)))
        return kp.value, ns.value, bool(potts.value)

    def SweepKernelName(self):
        """The sweep kernel's instance as a kernel trace names it."""
        buf = C.create_string_buffer(160)
        self._ck(lib.admm_sweep_kernel_name(self._h, buf, 160))
        return buf.value.decode()

    def LaunchInfo(self):
        """The options the last Compute launch ran with: {name: value}."""
        buf = C.create_string_buffer(512)
        self._ck(lib.admm_launch_info(self._h, buf, 512))
        out = {}
        for item in buf.value.decode().split():
            n, _, v = item.partition("=")
            out[n] = int(v) if v.lstrip("-").isdigit() else v
        return out

    def close(self):
        if self._h:
            lib.admm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise AdmmError(rc, (lib.admm_last_error(self._h) or b"").decode())


def edge_plan(edge_list, nranks, rank):
    """The edge solver's exchange plan for one rank (GetVerts + InitNeighborsByEdge,
    Grid.cpp:815-856, 198-331); host arithmetic only, no device.  Returns a dict:
    edge0, nedges, verts, counts, neighbours (ranks, ascending), and per neighbour the
    shared local vertex ids and that neighbour's copy counts."""
    el = np.ascontiguousarray(edge_list, dtype=np.int32).reshape(-1)
    info = np.zeros(5, dtype=np.int32)
    null = C.POINTER(C.c_int32)()
    p = lambda a: a.ctypes.data_as(_ip)
    rc = lib.admm_edge_plan(p(el), el.size // 2, nranks, rank, p(info), null, null, null, null, null, null)
    if rc != 0:
        raise AdmmError(rc, (lib.admm_last_error(None) or b"").decode())
    e0, ne, nv, nn, ns = (int(x) for x in info)
    verts, counts = np.zeros(max(nv, 1), np.int32), np.zeros(max(nv, 1), np.int32)
    nr, no = np.zeros(max(nn, 1), np.int32), np.zeros(nn + 1, np.int32)
    sl, sc = np.zeros(max(ns, 1), np.int32), np.zeros(max(ns, 1), np.int32)
    lib.admm_edge_plan(p(el), el.size // 2, nranks, rank, p(info), p(verts), p(counts), p(nr), p(no), p(sl), p(sc))
    return {"edge0": e0, "nedges": ne, "verts": verts[:nv], "counts": counts[:nv], "neighbours": nr[:nn],
            "shared": [sl[no[q]:no[q + 1]] for q in range(nn)], "shared_counts": [sc[no[q]:no[q + 1]] for q in range(nn)]}


def communicate_group(grids):
    """Label exchange between the strips of one problem held by this process
    (Grid objects in rank order)."""
    arr = (C.c_void_p * len(grids))(*[g._h for g in grids])
    rc = lib.admm_communicate_group(arr, len(grids))
    if rc != 0:
        raise AdmmError(rc, (lib.admm_last_error(grids[0]._h) or b"").decode())


def unique_id():
    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
    rc = lib.admm_comm_get_unique_id(buf)
    if rc != 0:
        raise AdmmError(rc, (lib.admm_last_error(None) or b"").decode())
    return bytes(buf.raw)
