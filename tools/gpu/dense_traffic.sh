# Where do the extra fetched lines of the dense 32-label sweep at four wavefronts per SIMD come from?
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; out=gpurun_out/r3/dense_traffic; mkdir -p $out
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  ADMM_DIRECT_LABELS=$1 ADMM_SWEEP_PAIRXCD=$2 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_$1_$2 -o vm --output-format csv -- python3 bench.py --workload 2048x2048_k32_dense --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --spin-up 0 > $out/pmc_$1_$2.log 2>&1 || exit $?
  python3 - $out/pmc_$1_$2 "direct=$1 pairxcd=$2" <<'PY'
import csv,glob,sys,collections
per=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE" and "chain_sweep" in r["Kernel_Name"] and "true, false, true, false" in r["Kernel_Name"]:
            per[r["Kernel_Name"][:70]][r["Dispatch_Id"]]=per[r["Kernel_Name"][:70]].get(r["Dispatch_Id"],0)+float(r["Counter_Value"])
for k,v in per.items(): print(sys.argv[2], "reads GB/launch", [round(x*2048/1e9,4) for x in v.values()])
PY
done
