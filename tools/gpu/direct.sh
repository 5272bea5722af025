# direct-label / deferred-sum tests, then the whole suite, then bench lines
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_direct_labels.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/direct_pytest.log 2>&1
rc=$?; tail -15 gpurun_out/direct_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r2_pytest.log; [ $rc -ne 0 ] && exit $rc
bash tools/bench_all.sh > gpurun_out/bench_all_direct.log 2>&1; rc=$?; cat gpurun_out/bench_all_direct.log; [ $rc -ne 0 ] && exit $rc
echo "== transposed labels, launched sums"
for w in 2048x2048_k32_potts 512x512_k16_trunc_linear 1024x1024_k64_dense "256x(256x256)_k8_potts" 64x64_k4_potts; do
ADMM_DIRECT_LABELS=0 ADMM_DEFER_SUMS=0 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --warmup 20 --workload "$w" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%s: %.1f it/s, %.4f ms per iteration, kernel %.4f ms' % (d['config']['workload'], d['value'], d['ms_per_step'], r['ms_per_launch']))" || exit 1
done
