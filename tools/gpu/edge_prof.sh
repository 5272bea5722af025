#!/bin/bash
# Edge solver: profiles of the three edge workloads, then the wait counters of the 32-label solve.
set -e
bash tools/profile_round.sh r02e edge_64x64_k4_potts edge_512x512_k16_trunc_linear edge_512x512_k32_potts
bash tools/gpu/pmc_wait.sh edge_512x512_k32_potts edge_wait
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d gpurun_out/edge_insts -o vm --output-format csv -- python bench.py --workload edge_512x512_k32_potts --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/edge_insts.log 2>&1
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
