# quick check on the GPU box: the Potts and parity tests, then bench lines
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_trunc.py tests/test_gpu_potts.py tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_regressions.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/quick_pytest.log 2>&1
rc=$?; tail -15 gpurun_out/quick_pytest.log; [ $rc -ne 0 ] && exit $rc
for w in 2048x2048_k32_potts "256x(256x256)_k8_potts" 1024x1024_k64_dense 512x512_k16_trunc_linear; do
timeout -k 10 120 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --workload "$w" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%s: %.1f it/s, %.4f ms per iteration, kernel %.4f ms, %.0f GB/s = %.1f%% of peak (%s)' % (d['config']['workload'], d['value'], d['ms_per_step'], r['ms_per_launch'], r['achieved'], 100*r['frac'], r.get('achieved_from','')))" || exit 1
done
timeout -k 10 200 python tools/strip_timing.py
