set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_pytest1.log 2>&1
rc=$?
tail -5 gpurun_out/r2_pytest1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2_noalloc.log 2>&1
rc=$?
tail -5 gpurun_out/r2_noalloc.log
exit $rc
