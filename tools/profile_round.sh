#!/bin/bash
# Runs on the GPU box (through gpurun): for every bench workload one bench line,
# the kernel trace of the same command, and the two HBM counter passes (separate
# runs: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Everything lands under
# gpurun_out/$1/; tools/distill_profiles.py turns it into the files kept under
# profiles/.
#   usage: bash tools/profile_round.sh r02 [workload ...]
set -u
tag=${1:-r02}
shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
step() { echo "== $*"; timeout -k 10 "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "step failed rc=$rc"; exit $rc; fi; }

if [ $# -gt 0 ]; then workloads=("$@"); else
workloads=(2048x2048_k32_potts 2048x2048_k32_dense 2048x2048_k32_potts_penalty0.3 1024x1024_k64_dense 512x512_k16_trunc_linear "256x(256x256)_k8_potts" 64x64_k4_potts edge_64x64_k4_potts edge_512x512_k16_trunc_linear edge_512x512_k32_potts)
fi
for w in "${workloads[@]}"; do
  n=$(echo "$w" | tr -d '()')
  extra="--no-cpu-baseline --no-secondary"
  [ "$w" = 2048x2048_k32_potts ] && extra=""
  # <workload>_penalty<p>: the workload with another ADMM penalty (binary32 multipliers unless p / 2 is a power of two)
  case "$w" in *_penalty*) extra="$extra --penalty ${w##*_penalty}"; w="${w%%_penalty*}";; esac
  echo "== bench $w"
  timeout -k 10 400 python bench.py --workload "$w" --steps 50 --warmup 5 $extra > "$out/bench_$n.json" 2> "$out/bench_$n.err" || { echo "bench $w failed"; tail -5 "$out/bench_$n.err"; exit 1; }
  step 300 rocprofv3 --kernel-trace --stats -d "$out/trace_$n" -o vm --output-format csv -- python bench.py --workload "$w" --steps 50 --warmup 5 --no-cpu-baseline --no-secondary ${extra##*--no-secondary} > "$out/trace_$n.log" 2>&1
  step 120 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch_$n" -o vm --output-format csv -- python bench.py --workload "$w" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ${extra##*--no-secondary} > "$out/pmc_fetch_$n.log" 2>&1
  step 120 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write_$n" -o vm --output-format csv -- python bench.py --workload "$w" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ${extra##*--no-secondary} > "$out/pmc_write_$n.log" 2>&1
done
echo done
