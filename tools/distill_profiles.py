#!/usr/bin/env python3
"""Turns the raw output of tools/profile_round.sh (gpurun_out/<tag>/) into the
small files kept under profiles/: per workload the bench line, the kernel
statistics of the same command, and the HBM traffic of its dominant kernel per
launch.

    python tools/distill_profiles.py r02

HBM bytes: FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950
FETCH_SIZE counts 64 B for each 128-B read request, so reads are doubled.  The
factor holds for every access width the sweep uses (the L2 fetches whole
128-byte lines whatever part of a line a wavefront asks for:
tools/fetch_calibration.hip, profiles/r03_fetch_calibration.json), and it is
checked again in the same pass on a kernel whose read volume is known
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
sys.path.insert(0, ROOT)


def counter(src, sub, name):
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                per.setdefault(r["Kernel_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
        n = os.path.basename(f)[len("bench_"):-len(".json")]
        line = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if not line:
            continue
        bench = json.loads(line[-1])
        open(os.path.join(dst, "%s_bench_%s.json" % (tag, n)), "w").write(line[-1] + "\n")
        stats = glob.glob(os.path.join(src, "trace_" + n, "**", "*kernel_stats.csv"), recursive=True)
        dominant = None
        if stats:
            shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, n)))
            rows = list(csv.DictReader(open(stats[0])))
            rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
            dominant = rows[0]
        fetch, write = counter(src, "pmc_fetch_" + n, "FETCH_SIZE"), counter(src, "pmc_write_" + n, "WRITE_SIZE")
        want = bench["roofline"]["kernel"]
        fk = [k for k in fetch if want in k]
        wk = [k for k in write if want in k]
        if not (fk and wk):
            print("%s: no counter pass names %s" % (n, want))
            continue
        cfg = bench["config"]
        V = cfg["rows"] * cfg["cols"] * cfg["batch"]
        check = {}
        for kn, v in fetch.items():
            if "chain_init_theta_kernel" in kn:
                known = V * cfg["k"] * 4
                check = {"kernel": "chain_init_theta_kernel", "bytes_read_known": known,
                         "FETCH_SIZE_KiB": v, "implied_factor": known / (v * 1024.0)}
        rd = fetch[fk[0]] * 1024.0 * 2.0
        wr = write[wk[0]] * 1024.0
        out = {
            "command": "rocprofv3 --pmc FETCH_SIZE -- python bench.py --workload '%s' --steps 3 --warmup 1 --no-cpu-baseline ; "
                       "the same with --pmc WRITE_SIZE (separate passes)" % cfg["workload"],
            "kernel": fk[0], "workload": cfg["workload"] + ", 1 GPU", "launch": cfg.get("launch"),
            "FETCH_SIZE_KiB_per_launch": fetch[fk[0]], "WRITE_SIZE_KiB_per_launch": write[wk[0]],
            "fetch_correction": 2.0, "fetch_correction_check": check,
            "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
            "hbm_bytes_per_launch": rd + wr,
            "stored_bytes_model": bench["roofline"].get("stored_bytes_model"),
            "algorithmic_bytes_per_launch": bench["roofline"].get("algorithmic_bytes", bench["roofline"].get("bytes_per_launch")),
        }
        if dominant is not None:
            out["kernel_trace"] = {"name": dominant["Name"], "calls": int(dominant["Calls"]),
                                   "average_ns": float(dominant["AverageNs"]), "percentage": float(dominant["Percentage"])}
            out["hbm_gbps_at_trace_duration"] = (rd + wr) / float(dominant["AverageNs"])
        json.dump(out, open(os.path.join(dst, "%s_hbm_traffic_%s.json" % (tag, n)), "w"), indent=1)
        print(n, json.dumps(out))


if __name__ == "__main__":
    main()
