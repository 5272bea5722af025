#!/bin/bash
# A/B of builds of the library on every chain workload of bench.py:
#   bash tools/gpu/ab_all.sh <name> [<name> ...]     (tools/gpu/lib_<name>.so, or "tree")
set -e
cd "${GRAFT_REPO_ROOT:-.}"
cp admm_amd/libadmm_hip.so /tmp/tree_libadmm_hip.so
run() { timeout -k 10 200 python3 bench.py --workload "$2" --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', '%.1f it/s' % d['value'], 'kernel %.4f ms' % d['roofline']['ms_per_launch'], d.get('objective_check',{}).get('ok'))"; }
for w in 2048x2048_k32_potts 2048x2048_k32_dense 1024x1024_k64_dense 512x512_k16_trunc_linear "256x(256x256)_k8_potts" 64x64_k4_potts; do
for i in 1 2; do
  for n in "$@"; do
    if [ "$n" = tree ]; then cp /tmp/tree_libadmm_hip.so admm_amd/libadmm_hip.so; else cp tools/gpu/lib_$n.so admm_amd/libadmm_hip.so; fi
    run $n "$w"
  done
done
done
cp /tmp/tree_libadmm_hip.so admm_amd/libadmm_hip.so
