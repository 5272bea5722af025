#!/bin/bash
# A workload (default: the headline) under each of the library's tuning knobs, against its own choices (same box).
#   bash tools/gpu/knobs.sh [workload]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
w=${1:-2048x2048_k32_potts}
run() { env "$@" timeout -k 10 200 python3 bench.py --workload "$w" --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.1f it/s' % d['value'], 'kernel %.4f ms' % d['roofline']['ms_per_launch'], d.get('objective_check',{}).get('ok'))"; }
for i in 1 2; do
run A=0
run ADMM_SWEEP_PCOMP=0
run ADMM_DIRECT_LABELS=1
run ADMM_SWEEP_XCD=1
run ADMM_SWEEP_NOMIX=1
run ADMM_SWEEP_PAIRXCD=0
run ADMM_DEFER_SUMS=0
run ADMM_LAM8=0
run ADMM_DIRECT_LABELS=0
run ADMM_SWEEP_XCD=0
done
