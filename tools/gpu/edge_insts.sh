#!/bin/bash
# Instruction counts of the 32-label edge solve (deterministic, unlike the rate from box to box),
# and its duration in the same process.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d gpurun_out/edge_insts -o vm --output-format csv -- python bench.py --workload edge_512x512_k32_potts --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/edge_insts.log 2>&1
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/edge_insts/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in per.items():
    if "edge_solve" in k:
        for c, vals in sorted(v.items()):
            print("   %-24s %.4g" % (c, sum(vals) / len(vals)))
PY
for i in 1 2 3; do
timeout -k 10 200 python bench.py --workload edge_512x512_k32_potts --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
