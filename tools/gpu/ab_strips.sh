#!/bin/bash
# A/B of builds of the library (tools/gpu/lib_<name>.so, or "tree") on the strip timings:
#   bash tools/gpu/ab_strips.sh <name> [<name> ...]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
cp admm_amd/libadmm_hip.so /tmp/tree_libadmm_hip.so
for i in 1 2; do
  for n in "$@"; do
    if [ "$n" = tree ]; then cp /tmp/tree_libadmm_hip.so admm_amd/libadmm_hip.so; else cp tools/gpu/lib_$n.so admm_amd/libadmm_hip.so; fi
    echo "== $n"
    timeout -k 10 200 python3 tools/strip_timing.py 4 8 16
  done
done
cp /tmp/tree_libadmm_hip.so admm_amd/libadmm_hip.so
