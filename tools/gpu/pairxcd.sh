# A/B of ADMM_SWEEP_PAIRXCD on the headline workload: rates alternating, then one FETCH_SIZE pass each.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r3/pairxcd; mkdir -p $out
for rep in 1 2 3; do
  for v in 0 1; do
    ADMM_SWEEP_PAIRXCD=$v timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $out/b_${v}_$rep.json 2> $out/b_${v}_$rep.err || exit $?
    python3 - $out/b_${v}_$rep.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("pairxcd", sys.argv[2], "it/s %.1f" % d["value"], "kernel ms %.4f" % d["roofline"]["ms_per_launch"], "check", d["objective_check"].get("ok"), d["objective_check"].get("labels_bit_exact"))
PY
  done
done
for v in 0 1; do
  ADMM_SWEEP_PAIRXCD=$v timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_$v -o vm --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $out/pmc_$v.log 2>&1 || exit $?
  python3 - $out/pmc_$v <<'PY'
import csv,glob,sys,collections
per=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE" and "chain_sweep" in r["Kernel_Name"]:
            per[r["Kernel_Name"][:60]][r["Dispatch_Id"]]=per[r["Kernel_Name"][:60]].get(r["Dispatch_Id"],0)+float(r["Counter_Value"])
for k,v in per.items(): print(sys.argv[1], k, "reads GB/launch", [round(x*2048/1e9,4) for x in v.values()])
PY
done
