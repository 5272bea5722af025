#!/bin/bash
# Every bench.py workload once (no CPU baseline), then the strip table.
set -e
cd "$(dirname "$0")/.."
for wl in 2048x2048_k32_potts 2048x2048_k32_dense 512x512_k16_trunc_linear 1024x1024_k64_dense "256x(256x256)_k8_potts" 64x64_k4_potts \
          edge_512x512_k32_potts edge_512x512_k16_trunc_linear edge_64x64_k4_potts; do
  echo "== $wl"
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 20 --workload "$wl" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.1f it/s, %.4f ms per iteration, kernel %.4f ms, %.0f GB/s = %.1f%% of peak (%s), stored-bytes model at %.1f%%' % (d['value'], d['ms_per_step'], r['ms_per_launch'], r['achieved'], 100*r['frac'], r.get('achieved_from','algorithmic'), 100*r.get('frac_of_model', 0)))"
done
echo "== 2048x2048_k32_potts --penalty 0.3 (binary32 multipliers)"
timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 20 --penalty 0.3 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s, kernel %.4f ms' % (d['value'], d['roofline']['ms_per_launch']))"
echo "== strips (strip 0 of G alone)"; timeout -k 10 200 python tools/strip_timing.py
echo "== strips (all G strips in one process, exchange included)"; timeout -k 10 200 python tools/strip_timing.py --group 2 4 8
