set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
run() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ.get('ADMM_DIRECT_LABELS','-'), '$*', '%.1f it/s' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel %.4f ms' % d['roofline']['ms_per_launch'], d['objective_check'].get('ok'))"; }
for rep in 1 2; do
  for dl in 0 1; do
    export ADMM_DIRECT_LABELS=$dl
    run --workload 2048x2048_k32_dense --steps 50 --warmup 5
    run --workload 1024x1024_k64_dense --steps 50 --warmup 5
  done
done
