#!/usr/bin/env python3
"""Turns the raw output of tools/profile_round.sh (gpurun_out/<tag>/) into the
small files kept under profiles/: bench lines, the kernel statistics of the
default bench command, and the HBM traffic of the sweep kernel per launch.

    python tools/distill_profiles.py r01

HBM bytes: FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950
FETCH_SIZE counts 64 B for each 128-B read request, so reads are doubled; the
factor is checked in the same pass on kernels whose read volume is known
(chain_init_theta_kernel reads its array exactly once).  WRITE_SIZE needs no
correction.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEP = "chain_sweep_kernel<32, 1, 8, 4, false, true, true>"
WORKLOAD = "2048x2048_k32_potts"


def counter(tag, sub, name):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", tag, sub, "**", "*counter_collection.csv"), recursive=True)
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                per.setdefault(r["Kernel_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, "bench_*.json")):
        name = os.path.basename(f).replace("(", "").replace(")", "")
        line = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if line:
            json.loads(line[-1])
            open(os.path.join(dst, "%s_%s" % (tag, name)), "w").write(line[-1] + "\n")
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, WORKLOAD)))
    fetch, write = counter(tag, "pmc_fetch", "FETCH_SIZE"), counter(tag, "pmc_write", "WRITE_SIZE")
    fk = [k for k in fetch if SWEEP in k]
    wk = [k for k in write if SWEEP in k]
    if fk and wk:
        rd = fetch[fk[0]] * 1024.0 * 2.0
        wr = write[wk[0]] * 1024.0
        V, k = 2048 * 2048, 32
        check = {}
        for kn, v in fetch.items():
            if "chain_init_theta_kernel" in kn:
                check = {"kernel": "chain_init_theta_kernel", "bytes_read_known": V * k * 4,
                         "FETCH_SIZE_KiB": v, "implied_factor": V * k * 4 / (v * 1024.0)}
        out = {
            "command": "rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline ; "
                       "the same with --pmc WRITE_SIZE (separate passes)",
            "kernel": "admm::" + SWEEP, "workload": WORKLOAD + ", 1 GPU",
            "FETCH_SIZE_KiB_per_launch": fetch[fk[0]], "WRITE_SIZE_KiB_per_launch": write[wk[0]],
            "fetch_correction": 2.0, "fetch_correction_check": check,
            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
            "hbm_bytes_per_launch": rd + wr,
            "algorithmic_bytes_per_launch": V * (32 * k + 16),
        }
        json.dump(out, open(os.path.join(dst, "%s_hbm_traffic_sweep.json" % tag), "w"), indent=1)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
