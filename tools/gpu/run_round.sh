# GPU-box session: whole GPU suite, the two-process launch of the bench on one GPU
# (RCCL refuses it: the run must fall back to the host transport and exit 0), the
# memory-pattern table, and the strip table.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_pytest2.log 2>&1
rc=$?; tail -3 gpurun_out/r2_pytest2.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r2_two_ranks.json 2> gpurun_out/r2_two_ranks.err
rc=$?; echo "two ranks on one GPU: exit $rc"; tail -3 gpurun_out/r2_two_ranks.err; cat gpurun_out/r2_two_ranks.json; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 tools/access_patterns > gpurun_out/r2_access.log 2>&1
rc=$?; cat gpurun_out/r2_access.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 bash tools/bench_all.sh > gpurun_out/r2_bench_all.log 2>&1
rc=$?; cat gpurun_out/r2_bench_all.log; exit $rc
