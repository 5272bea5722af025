# truncated-table kernel: its tests, then its sweep time beside the general kernel's
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_trunc.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/trunc_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/trunc_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/trunc_compare.py
