# In-process strips: tests of the exchange, then its cost with and without the pull.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests/test_gpu_regressions.py tests/test_gpu_potts.py tests/test_gpu_random.py tests/test_gpu_trunc.py tests/test_gpu_host_driver.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3/pull_pytest.log 2>&1
rc=$?; tail -4 gpurun_out/r3/pull_pytest.log; [ $rc -ne 0 ] && exit $rc
for v in 0 1 0 1; do
  echo "ADMM_GROUP_PULL=$v"
  ADMM_GROUP_PULL=$v timeout -k 10 300 python3 tools/strip_timing.py --group 2 4 8 || exit $?
done
timeout -k 10 300 python3 tools/strip_timing.py 1 2 4 8
