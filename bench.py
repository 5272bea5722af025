#!/usr/bin/env python3
"""Benchmark of the ADMM hot path: row/column chain min-sum on a 2048 x 2048 grid, k = 32.

    python bench.py --gpus N --steps K --warmup W

One step is one ADMM iteration as the reference's driver loop runs it
(admm.cpp:78-100): Compute, Communicate, Update and the per-iteration linear
objective.  The objective values are left in device slots and read after the
timed region, so the loop never waits for the host.  With N > 1 the same grid is
cut into N row strips / column strips (strong scaling), one process per GPU, and
the chain labels cross GPUs through RCCL inside admm_communicate.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _requested_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def _self_launch():
    """`python bench.py --gpus N` typed as it stands (the reference's equivalent is one
    `mpiexec -n N` line, admm.cpp:27): start one process per GPU through
    torch.distributed.run as CHILD processes and leave with the launcher's exit code.
    This runs before the HIP library is loaded, so the parent never touches a GPU."""
    import socket
    import subprocess
    n = _requested_gpus(sys.argv[1:])
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _requested_gpus(sys.argv[1:]) > 1:
    _self_launch()

# The HIP library comes first so that the process uses one HIP / RCCL runtime,
# the one the library was built against.
import admm_amd as A  # noqa: E402
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, datasheet

WORKLOADS = {
    # name: rows, cols, k, batch, pairwise
    "2048x2048_k32_potts": (2048, 2048, 32, 1, "potts"),
    "512x512_k16_trunc_linear": (512, 512, 16, 1, "trunc_linear"),
    "1024x1024_k64_dense": (1024, 1024, 64, 1, "dense"),
    # the headline grid with an arbitrary k x k table: the general min-plus kernel
    "2048x2048_k32_dense": (2048, 2048, 32, 1, "dense"),
    "256x(256x256)_k8_potts": (256, 256, 8, 256, "potts"),
    "64x64_k4_potts": (64, 64, 4, 1, "potts"),
    # the reference's own algorithm (one tree per edge, Tree.cpp:112-274): secondary
    "edge_64x64_k4_potts": (64, 64, 4, 1, "potts"),
    "edge_512x512_k16_trunc_linear": (512, 512, 16, 1, "trunc_linear"),
    "edge_512x512_k32_potts": (512, 512, 32, 1, "potts"),
}


def bytes_per_iteration(V, k):
    """Algorithmic HBM bytes of one ADMM iteration of the chain solver, float32 operands
    and int32 labels: V * (36 k + 24)."""
    return V * (36 * k + 24)


def sweep_bytes(V, k):
    """Share of the chain sweep kernel in that figure: theta read twice (8k), both lambda
    arrays read and written (16k), consensus read twice (8k), both label arrays read and
    written (16)."""
    return V * (32 * k + 16)


def stored_bytes(V, k, kp, chains_per_wave, potts, lam_el, lam_shared, pcomp=False, pair_xcd=True):
    """Bytes one sweep launch has to move given how the operands are stored: a lower bound
    for this representation (every line touched once per use, nothing re-read).
      theta       4k per node, read by the row chains and by the column chains
      multipliers one array on one device (read by both sets, written once) or one per set
                  (each read and written); lam_el bytes per entry: 4 as binary32, 2 or 1 as counts.
                  The memory system moves whole 128-byte lines: where the column chains of one
                  wavefront use less than a line per step (2 x 32 int8 counts = 64 bytes) and the
                  wavefront that uses the rest runs on another die (pair_xcd off), the column
                  chains' read costs a full line per wavefront
      labels      1 byte: every set reads its own and the crossing chains', writes its own
      back-pointers  Potts step: two 64-bit lane masks per wavefront step (16 B for
                  chains_per_wave nodes), written by the forward pass and read by the
                  trace-back, plus with path compression one byte per lane and block of eight
                  steps, likewise; general step: one byte per padded label per node, likewise"""
    piece = chains_per_wave * kp * lam_el
    col_read = lam_el * k * (1.0 if (piece >= 128 or pair_xcd) else 128.0 / piece)
    lam = (lam_el * k * 2 if lam_shared else lam_el * k * 3) + col_read
    labels = 2 * 3
    bp = 2 * 2 * (16.0 / chains_per_wave if potts else kp)
    if potts and pcomp:
        bp += 2 * 2 * (64.0 / 8 / chains_per_wave)
    return V * (8 * k + lam + labels + bp)


def edge_bytes(V, E, k):
    """Algorithmic bytes of one iteration of the edge solver: per edge mu_e read + written and
    theta_e read (3 k^2), the six 2k node vectors of class Edge read or written once or twice
    (14 k), plus the consensus row of every vertex."""
    return 4 * (E * (3 * k * k + 14 * k) + V * k)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(rows, cols, k, pairwise, seed, batch=1):
    """The same iteration on the host cores, as bounded samples (about 20 s in all):
    the CPU restatement of the chain solver with one chain per thread (the headline
    `value`) and on one thread, and the reference's own Tree.cpp (built into oracle/_ref
    where its sources were present) on a small grid of the same k, since the reference has
    only the per-edge solver."""
    from oracle import oracle as O
    ncpu = os.cpu_count() or 1
    threads = min(16, ncpu)
    V1 = rows * cols
    un, W = O.synth_problem(rows, cols, k, seed, pairwise)
    out = {"unit": "it/s", "kind": "port", "nproc": ncpu, "cpu_model": cpu_model()}

    # all threads: whole iterations of one grid (a batch is `batch` such grids in a row)
    O.set_threads(threads)
    o = O.ChainOracle(rows, cols, k, un, W)
    o.init()
    t0 = time.perf_counter()
    its = 0
    while True:
        o.compute(); o.update(); o.objective(dual=False)
        its += 1
        if time.perf_counter() - t0 > 8.0 or its >= 20:
            break
    dt = time.perf_counter() - t0
    out["value"] = its / dt / batch
    out["cores"] = threads
    out["sample"] = ("%d full ADMM iterations (compute + update + linear objective) of one %dx%d k=%d grid%s, "
                     "oracle/admm_oracle.c with one chain per OpenMP thread"
                     % (its, rows, cols, k, ", divided by the %d grids of the batch" % batch if batch > 1 else ""))

    # one thread: a slice of the chains of one iteration, scaled to all of them
    O.set_threads(1)
    nr = max(1, min(rows, int(rows * 4.0 / max(1.0, dt / its * threads))))  # about 4 s
    nc = max(1, min(cols, int(cols * nr / rows)))
    t0 = time.perf_counter()
    o.compute(0, nr, 0, nc)
    d1 = time.perf_counter() - t0
    frac = (nr * cols + nc * rows) / (2.0 * V1)
    out["single_thread"] = {
        "value": frac / d1 / batch, "unit": "it/s", "cores": 1,
        "sample": "Compute of %d of %d row chains and %d of %d column chains, scaled to all chains" % (nr, rows, nc, cols)}
    del o

    # the reference's own solver (per-edge Bethe sum-product) at this k, one core
    ref = O.load_ref()
    R2 = C2 = 64 if k <= 16 else 32
    un2, W2 = O.synth_problem(R2, C2, k, seed, pairwise)
    el = O.grid_edges(R2, C2)
    E2 = el.size // 2
    ep = np.tile(W2.reshape(-1), E2)
    e = O.RefEdge(ref, R2 * C2, el, k, un2, ep) if ref is not None else O.EdgeOracle(R2 * C2, el, k, un2, ep)
    e.init()
    t0 = time.perf_counter()
    n = 0
    while True:
        e.compute(); e.update(); e.objective()
        n += 1
        if time.perf_counter() - t0 > 4.0 or n >= 50:
            break
    de = (time.perf_counter() - t0) / n
    E1 = rows * (cols - 1) + cols * (rows - 1)
    out["edge_solver"] = {
        "kind": "reference" if ref is not None else "port", "cores": 1,
        "us_per_edge_iteration": 1e6 * de / E2,
        "value": 1.0 / (de / E2 * E1) / batch, "unit": "it/s",
        "sample": "%d iterations of a %dx%d k=%d grid (%d edges) through %s, scaled by edges to the %dx%d grid"
                  % (n, R2, C2, k, E2, "the reference's Tree.cpp (oracle/_ref)" if ref is not None else "oracle/admm_oracle.c", rows, cols)}
    return out


def measured_traffic(workload, world, kernel, launch=None):
    """HBM bytes per launch of the dominant kernel from the committed counter passes
    (FETCH_SIZE and WRITE_SIZE come from rocprofv3 --pmc runs of this same command; they
    cannot be collected from inside this process).  Only a pass of the same workload and
    the same kernel instance counts; otherwise None."""
    import glob
    if world != 1:
        return None, None
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "r*_hbm_traffic_*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if str(d.get("workload", "")).startswith(workload) and kernel in str(d.get("kernel", "")):
            if launch is not None and d.get("launch") is not None and d.get("launch") != launch:
                continue  # a pass of the same kernel under other launch options moves other bytes
            return d.get("hbm_bytes_per_launch"), "profiles/" + os.path.basename(f)
    return None, None


GOLDEN = os.path.join(ROOT, "tests", "golden", "bench_objectives.json")


def golden_name(workload, penalty):
    return workload if penalty == 0.5 else "%s_penalty%g" % (workload, penalty)


def objective_check(workload, penalty, seed, first_it, linear, total_its=None, final=None, labels_sha=None):
    """What the run left behind against the committed values of the same trajectory
    (tests/golden/bench_objectives.json, written in the build container by
    tests/golden/make_bench_objectives.py with the CPU oracle; nothing of the oracle runs
    here).  linear[i] is the linear objective after iteration first_it + i (1-based);
    final = (primal, linear, dual) after total_its iterations."""
    try:
        doc = json.load(open(GOLDEN))
    except (OSError, ValueError):
        return {"checked": False, "why": "tests/golden/bench_objectives.json is missing"}
    case = doc.get(golden_name(workload, penalty))
    if case is None or seed != case.get("seed"):
        return {"checked": False, "why": "no committed trajectory for this workload, penalty and seed"}
    gl = case["linear"]
    n = max(0, min(len(linear), len(gl) - (first_it - 1)))
    out = {"checked": n > 0, "source": "tests/golden/bench_objectives.json", "iterations_compared": n,
           "first_iteration": first_it}
    if n > 0:
        exp = np.array(gl[first_it - 1:first_it - 1 + n])
        got = np.array(linear[:n], dtype=np.float64)
        rel = np.abs(got - exp) / np.abs(exp)
        out["linear"] = {"expected_last": float(exp[-1]), "got_last": float(got[-1]), "rel_err": float(rel.max())}
        out["rel_err"] = float(rel.max())
    at = case["at"].get(str(total_its)) if total_its is not None else None
    if at is not None and final is not None:
        for name, got in zip(("primal", "linear", "dual"), final):
            e = at[name]
            r = abs(got - e) / abs(e)
            out[name + "_at_%d" % total_its] = {"expected": e, "got": float(got), "rel_err": r}
            out["rel_err"] = max(out.get("rel_err", 0.0), r)
        if labels_sha is not None:
            out["labels_bit_exact"] = (labels_sha == at["labels_sha256"])
        out["checked"] = True
    if out["checked"]:
        out["ok"] = out["rel_err"] <= 1e-9 and out.get("labels_bit_exact", True)
    return out


def labels_sha256(g):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(g.Labels(), dtype="<i4").tobytes()).hexdigest()


def quick_rate(workload, penalty, seed, steps=12, warmup=40):
    """A short single-GPU run of another workload, for the `secondary` fields of the
    headline line: the same step as the timed region of main(), after enough untimed
    iterations for the GPU to reach its sustained clocks (the host work of setting a problem
    up leaves it idle for half a second)."""
    rows, cols, k, batch, pairwise = WORKLOADS[workload]
    g = A.Grid(rank=0, nprocs=1, device=0)
    g.SetPartitioner(A.ROW)
    g.SetPenalties(penalty, 1.0)
    g.SetGlobalInfo(rows, cols, k, batch=batch)
    g.Generate(seed=seed, pairwise=pairwise)
    g.InitOptimization()
    g._unary = None
    for i in range(warmup):
        g.Compute(); g.Communicate(); g.Update(); g.ObjectiveEnqueue(1 + i)
    g.Synchronize()
    g.ProfileReset(3)
    t0 = time.perf_counter()
    for i in range(steps):
        g.Compute(); g.Communicate(); g.Update(); g.ObjectiveEnqueue(1 + warmup + i)
    g.Synchronize()
    dt = time.perf_counter() - t0
    kernel_ms, launches = g.ProfileRead()
    g.ProfileReset(False)
    linear = g.ObjectiveRead(1, warmup + steps)
    out = {"workload": workload, "penalty": penalty, "value": steps / dt, "unit": "it/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt / steps, "kernel": g.SweepKernelName(), "ms_per_launch": kernel_ms,
           "objective_check": objective_check(workload, penalty, seed, 1, linear)}
    g.FinalizeOptimization()
    g.close()
    return out


def make_grid(args, rank, nprocs, device, rows, cols, k, batch, pairwise, seed, edge, keep_host=False):
    g = A.Grid(rank=rank, nprocs=nprocs, device=device)
    g.SetPartitioner(A.EDGE if edge else A.ROW)
    g.SetPenalties(args.penalty, 1.0)  # admm.cpp:44: 0.5, 1.0
    g.SetGlobalInfo(rows, cols, k, batch=batch)
    g.exchange = 0 if args.exchange == "labels" else 1
    g.Generate(seed=seed, pairwise=pairwise)
    g.InitOptimization()
    if not keep_host:
        g._unary = None  # the host copy is no longer needed
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="2048x2048_k32_potts", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--penalty", type=float, default=0.5,
                    help="ADMM penalty (admm.cpp:44 sets 0.5).  Half of it a power of two lets the chain solver keep its "
                         "multipliers as integer counts; any other value runs the binary32 multiplier kernels")
    ap.add_argument("--spin-up", type=int, default=100,
                    help="iterations of a throw-away context of the same problem run right before the W warm-up steps "
                         "(0: none).  A GPU that has been idle runs its first tens of milliseconds below its sustained "
                         "clocks; the warm-up steps of a 0.4 ms iteration end long before that")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline workload on one GPU: skip the short runs of the general-table kernel and of a "
                         "penalty that is not a power of two, which are reported as `secondary`")
    ap.add_argument("--exchange", default="labels", choices=["labels", "dense"])
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "host"],
                    help="N > 1: RCCL send/recv inside admm_communicate, or labels staged through the host "
                         "and moved by gloo; auto tries RCCL and falls back to host if its bootstrap fails")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
        args.gpus = world
    if args.steps + args.warmup >= 4000:
        sys.exit("steps + warmup must stay below 4000 (objective slots)")

    dist = None
    if world > 1:
        import torch.distributed as dist  # control plane only (gloo): id broadcast, barrier, max
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    edge = args.workload.startswith("edge_")
    rows, cols, k, batch, pairwise = WORKLOADS[args.workload]
    my_batch = batch
    nprocs_for_grid = world
    if batch > 1 and world > 1:
        # independent grids: shard the batch, no exchange
        b0, my_batch = A.partition(batch, world, rank)
        nprocs_for_grid = 1
    else:
        b0 = 0

    ndev = A._lib.lib.admm_device_count()
    if ndev < 1:
        sys.exit("bench.py needs a HIP device (there is no CPU fallback)")
    device = local_rank % ndev
    grank = rank if nprocs_for_grid > 1 else 0
    g = make_grid(args, grank, nprocs_for_grid, device, rows, cols, k, my_batch, pairwise, args.seed + b0, edge,
                  keep_host=args.spin_up > 0)
    spin = None
    if args.spin_up > 0:
        # A throw-away context of the same problem (other memory, no part of any measurement),
        # fed from the host arrays the measured one was made of: see the spin-up before the warm-up steps
        spin = A.Grid(rank=0, nprocs=1, device=device)
        spin.SetPartitioner(A.EDGE if edge else A.ROW)
        spin.SetPenalties(args.penalty, 1.0)
        spin.SetGlobalInfo(rows, cols, k, batch=my_batch)
        spin.Load(g._unary, g._pairwise)
        spin.InitOptimization()
        spin._unary = None
        g._unary = None
    transport = "none"
    done_its = 0  # iterations this context has run before the warm-up (the RCCL probe)
    if nprocs_for_grid > 1:
        transport = args.transport
        if transport in ("auto", "rccl"):
            import torch
            ok = 1
            try:
                ids = [A.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                g.CommInit(ids[0])
            except A.AdmmError as e:
                ok = 0
                sys.stderr.write("rank %d: RCCL bootstrap failed: %s\n" % (rank, e))
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                # one untimed probe iteration: an exchange that fails on any rank
                # sends everybody to the host-staged transport instead of
                # leaving the others waiting at a barrier
                try:
                    g.Compute(); g.Communicate(); g.Update()
                    if not g.SynchronizeWithin(120.0):
                        # a partner that never answers: leave with an error instead of holding the
                        # node until the caller's limit (the queued exchange cannot be taken back)
                        sys.stderr.write("rank %d: the RCCL exchange did not complete within 120 s\n" % rank)
                        sys.stderr.flush()
                        os._exit(3)
                except A.AdmmError as e:
                    ok = 0
                    sys.stderr.write("rank %d: RCCL exchange failed: %s\n" % (rank, e))
                flag = torch.tensor([ok], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                transport = "rccl"
                done_its = 1
            elif transport == "rccl":
                sys.exit("RCCL failed and --transport rccl was requested")
            else:
                transport = "host"
                g.close()
                g = make_grid(args, grank, nprocs_for_grid, device, rows, cols, k, my_batch, pairwise, args.seed + b0, edge)

    def barrier():
        g.Synchronize()
        if dist is not None:
            dist.barrier()
        g.Synchronize()

    edge_obj = []

    def step(i):
        g.Compute()
        if transport == "host":
            g.CommunicateHost(dist)
        else:
            g.Communicate()
        g.Update()
        if edge:
            edge_obj.append(g.ComputeLinearObj())  # the reference's loop reads it every iteration (admm.cpp:94)
        else:
            g.ObjectiveEnqueue(1 + i)

    if spin is not None:
        # A GPU that has just sat idle while the host generated and uploaded the problem runs below
        # its sustained clocks for tens of milliseconds (the sweep's clock counter shows 1.6 GHz in the
        # first iterations of a process, 2.2 GHz fifty iterations later), and the W warm-up steps of a
        # 0.4 ms iteration are over long before that.  The throw-away context keeps it busy right
        # up to them; the measured context, its trajectory and the K timed steps are untouched.
        for _ in range(args.spin_up):
            spin.Compute(); spin.Communicate(); spin.Update()
        spin.Synchronize()
    for i in range(args.warmup):
        step(i)
    barrier()
    # HIP events around a sample of the Compute launches (one in four of a 20-step run, every
    # launch of a shorter one): the two records cost several microseconds of stream time each time
    g.ProfileReset(max(1, min(8, args.steps // 5)))
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    g.Synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    kernel_ms, launches = g.ProfileRead()
    g.ProfileReset(False)
    linear = None
    sha = None
    if edge:
        last = np.array([edge_obj[-1], edge_obj[-1], float("nan")], dtype=np.float64)
    else:
        linear = g.ObjectiveRead(1 + args.warmup, args.steps)
        primal, dual = g.Objective(dual=True)  # decoded energy and dual bound, evaluated once at the end
        last = np.array([primal, linear[-1], dual], dtype=np.float64)
        if world == 1 and batch * rows * cols <= (1 << 25):
            sha = labels_sha256(g)
    if dist is not None:
        import torch
        tt = torch.from_numpy(last)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)  # admm.cpp:95
        if linear is not None:
            tl = torch.from_numpy(linear)
            dist.all_reduce(tl, op=dist.ReduceOp.SUM)
        sm = torch.tensor([kernel_ms], dtype=torch.float64)
        dist.all_reduce(sm, op=dist.ReduceOp.MAX)
        kernel_ms = float(sm[0])

    V = rows * cols * batch
    E = (rows * (cols - 1) + cols * (rows - 1)) * batch
    it_per_s = args.steps / elapsed
    sec = kernel_ms * 1e-3

    out = None
    if rank == 0:
        config = {
            "workload": args.workload, "rows": rows, "cols": cols, "k": k, "batch": batch,
            "pairwise": pairwise, "solver": "edge_bethe" if edge else "chain_minsum", "penalty": args.penalty, "rho": 1.0,
            "partition": ("edge slices x%d" % world) if edge else ("row/column strips x%d" % world if batch == 1 else "batch sharded x%d" % world),
            "exchange": args.exchange, "transport": transport,
            "spin_up": "%d untimed iterations of a throw-away context of the same problem before the warm-up steps" % args.spin_up
                       if args.spin_up > 0 else "none",
        }
        if edge:
            b_it = edge_bytes(V, E, k)
            kernel = "edge_solve_kernel<%d>" % (1 << max(1, (k - 1).bit_length()))
            config["kernel"] = kernel
            config["launch"] = g.LaunchInfo()
            traffic, traffic_src = measured_traffic(args.workload, world, "edge_solve_kernel")
            # what the kernel is really bound by: binary64 exp / log on the vector ALU
            transc = E * (4 * k * k + 4 * k)
            roof = {
                "kernel": kernel, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "achieved": b_it / world / sec / 1e9 if sec > 0 else 0.0,
                "bytes_per_launch": b_it / world, "ms_per_launch": kernel_ms, "launches_timed": launches,
                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "transcendentals_per_launch": transc, "transcendentals_per_sec": transc / world / sec if sec > 0 else 0.0,
                "note": "achieved = algorithmic bytes 4*(E*(3k^2+14k)+V*k) over the kernel's duration.  The kernel is not "
                        "HBM bound: every edge costs 4k^2+4k binary64 exp/log calls (the reference's arithmetic, kept for "
                        "parity), and transcendentals_per_sec is the rate that limits it.",
            }
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        else:
            kp, ns, potts = g.SweepKernelInfo()
            kernel = g.SweepKernelName()
            flags = [f.strip(" >") for f in kernel.split(",")]
            lam16, lam8 = flags[6] == "true", len(flags) > 8 and flags[8] == "true"
            pcomp = len(flags) > 7 and flags[7] == "true"
            launch = g.LaunchInfo()
            config.update({"kernel": kernel, "potts_step": potts, "lam16": lam16, "lam8": lam8, "launch": launch})
            b_it = bytes_per_iteration(V, k)
            algo = sweep_bytes(V, k) / world
            model = stored_bytes(V, k, kp, 64 // (kp * ns), potts, 1 if lam8 else (2 if lam16 else 4), lam_shared=(nprocs_for_grid == 1),
                                 pcomp=pcomp, pair_xcd=bool(launch.get("pair_xcd", 0)) or bool(launch.get("xcd_remap", 0))) / world
            traffic, traffic_src = measured_traffic(args.workload, world, kernel, launch)
            moved = traffic if traffic else model
            roof = {
                "kernel": kernel, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # the rate the memory system ran at: bytes really moved over the kernel's duration
                "achieved": moved / sec / 1e9 if sec > 0 else 0.0,
                "achieved_from": "traffic" if traffic else "stored_bytes_model",
                "ms_per_launch": kernel_ms, "launches_timed": launches,
                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "stored_bytes_model": model, "frac_of_model": model / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0,
                "algorithmic_bytes": algo, "algorithmic_gbps": algo / sec / 1e9 if sec > 0 else 0.0,
                "note": "frac = bytes the kernel moves per launch (HBM counters of a rocprofv3 pass of this command when one is "
                        "committed for this kernel instance, else the stored-representation model) / kernel time / 8 TB/s.  "
                        "stored_bytes_model is the least this representation can move: theta twice, the multipliers (on one "
                        "device one array read twice and written once; int8 or int16 counts with a power-of-two penalty, "
                        "see config.lam8 / lam16, binary32 otherwise), byte labels, back-pointers.  algorithmic_bytes = V*(32k+16) prices the same launch "
                        "with float32 multipliers, a consensus array and int32 labels; the kernel moves fewer bytes than "
                        "that (same numbers, exact), so algorithmic_gbps may exceed the peak and is not a rate.",
            }
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        out = {
            "metric": "admm_iterations_per_sec",
            "value": it_per_s,
            "unit": "it/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if (batch > 1) else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": config,
            "label_updates_per_sec": None if edge else 2.0 * V * it_per_s,
            "iteration_bytes": b_it,
            "iteration_gbps": b_it * it_per_s / 1e9,
            "final_objective": {"primal": float(last[0]), "linear": float(last[1]), "dual": float(last[2])},
            "roofline": roof,
        }
        if linear is not None:
            total = done_its + args.warmup + args.steps
            out["objective_check"] = objective_check(args.workload, args.penalty, args.seed, done_its + args.warmup + 1,
                                                     linear, total, last, sha)
    g.FinalizeOptimization()
    g.close()
    if spin is not None:
        spin.close()
    if rank == 0:
        if world == 1 and args.workload == "2048x2048_k32_potts" and args.penalty == 0.5 and not args.no_secondary:
            # the same grid where the specialisations of the headline do not apply
            out["secondary"] = {
                "general_table": quick_rate("2048x2048_k32_dense", 0.5, args.seed),
                "binary32_multipliers": quick_rate("2048x2048_k32_potts", 0.3, args.seed),
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rows, cols, k, pairwise, args.seed, batch)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
