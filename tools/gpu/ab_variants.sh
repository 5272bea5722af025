#!/bin/bash
# A/B of several builds of the library on one box: tools/gpu/lib_<name>.so (experimental builds,
# `make OUT=... EXTRA=-D...`) and the tree's own (name "tree"), alternating, on one workload.
#   bash tools/gpu/ab_variants.sh <workload> <name> [<name> ...]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
w=$1; shift
cp admm_amd/libadmm_hip.so /tmp/tree_libadmm_hip.so
run() { timeout -k 10 200 python3 bench.py --workload "$w" --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.1f it/s' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms' % d['roofline']['ms_per_launch'])"; }
for i in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = tree ]; then cp /tmp/tree_libadmm_hip.so admm_amd/libadmm_hip.so; else cp tools/gpu/lib_$n.so admm_amd/libadmm_hip.so; fi
    run $n
  done
done
cp /tmp/tree_libadmm_hip.so admm_amd/libadmm_hip.so
