#!/bin/bash
# A/B of two builds of the library on one box: tools/gpu/prev_libadmm_hip.so (a build of an
# earlier commit, put there by hand) against the tree's, alternating, on one workload.
#   bash tools/gpu/ab_lib.sh <workload>
set -e
w=${1:-edge_512x512_k32_potts}
cp admm_amd/libadmm_hip.so /tmp/new_libadmm_hip.so
run() { timeout -k 10 200 python bench.py --workload "$w" --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  cp tools/gpu/prev_libadmm_hip.so admm_amd/libadmm_hip.so; run prev
  cp /tmp/new_libadmm_hip.so admm_amd/libadmm_hip.so; run new
done
