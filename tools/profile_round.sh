#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines of every workload, the
# kernel trace of the default bench command, and the two HBM counter passes.
# Everything lands under gpurun_out/$1/; tools/distill_profiles.py turns it
# into the files kept under profiles/.
#   usage: bash tools/profile_round.sh r01
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
step() { echo "== $*"; timeout -k 10 "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "step failed rc=$rc"; exit $rc; fi; }

for w in 2048x2048_k32_potts 1024x1024_k64_dense 512x512_k16_trunc_linear "256x(256x256)_k8_potts" 64x64_k4_potts; do
  extra=""
  [ "$w" != 2048x2048_k32_potts ] && extra="--no-cpu-baseline"
  echo "== bench $w"
  timeout -k 10 400 python bench.py --workload "$w" --steps 50 --warmup 5 $extra > "$out/bench_$w.json" 2> "$out/bench_$w.err" || { echo "bench $w failed"; exit 1; }
done
step 300 rocprofv3 --kernel-trace --stats -d "$out/trace" -o vm --output-format csv -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$out/trace.log" 2>&1
step 120 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o vm --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1
step 120 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o vm --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1
echo done
