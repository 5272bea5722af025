set -e
cd "${GRAFT_REPO_ROOT:-.}"
cp admm_amd/libadmm_hip.so /tmp/tree_libadmm_hip.so
run() { timeout -k 10 200 python3 bench.py --penalty 0.3 --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'penalty 0.3', '%.1f it/s' % d['value'], 'kernel %.4f ms' % d['roofline']['ms_per_launch'], d.get('objective_check',{}).get('ok'))"; }
for i in 1 2 3; do
  cp tools/gpu/lib_head.so admm_amd/libadmm_hip.so; run head
  cp /tmp/tree_libadmm_hip.so admm_amd/libadmm_hip.so; run tree
done
