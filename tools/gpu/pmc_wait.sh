# SQ counter pass of one bench workload: what the wavefronts wait for.
#   bash tools/gpu/pmc_wait.sh <workload> <tag>
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
w=$1; tag=$2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d gpurun_out/$tag -o vm --output-format csv -- python bench.py --workload "$w" --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag.log 2>&1 || { tail -5 gpurun_out/$tag.log; exit 1; }
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in per.items():
    if "sweep" in k or "edge_solve" in k:
        print(k[:100])
        for c, vals in sorted(v.items()):
            print("   %-24s %.4g" % (c, sum(vals) / len(vals)))
PY
