# Does a short run (the 25 iterations the round-end benchmark is run with) pay for clocks that have not ramped up yet?
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
run() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.1f it/s' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms over %d' % (d['roofline']['ms_per_launch'], d['roofline']['launches_timed']), d['objective_check'].get('ok'))"; }
for rep in 1 2 3; do
  run --steps 20 --warmup 5 --spin-up 0
  run --steps 20 --warmup 5 --spin-up 100
  run --steps 20 --warmup 5 --spin-up 60
  run --steps 50 --warmup 5 --spin-up 100
done
