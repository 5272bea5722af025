# General (dense table) kernels: parity tests, then rates.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_general_counts.py tests/test_gpu_regressions.py "tests/test_gpu_fullsize.py::test_bench_trajectory_25_iterations" tests/test_gpu_fullsize.py::test_tall_grid_far_offsets tests/test_gpu_fullsize.py::test_config_1024x1024_k64_dense -x -q --timeout 300 --timeout-method thread > gpurun_out/r3/dense_pytest.log 2>&1
rc=$?; tail -4 gpurun_out/r3/dense_pytest.log; [ $rc -ne 0 ] && exit $rc
run() { timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.1f it/s' % d['value'], 'kernel %.4f ms' % d['roofline']['ms_per_launch'], d['config']['kernel'], d['objective_check'].get('ok'))"; }
for rep in 1 2; do
  run --workload 2048x2048_k32_dense --steps 50 --warmup 5
  run --workload 2048x2048_k32_dense --steps 50 --warmup 5 --penalty 0.3
  run --workload 1024x1024_k64_dense --steps 50 --warmup 5
done
