# The whole GPU suite on the GPU box (through gpurun), log under gpurun_out/.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/r3_pytest.log
exit $rc
