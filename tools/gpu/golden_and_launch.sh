# The golden-pinned full-size tests, the bench line with its objective check, and the self-launching two-rank bench.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest \
  "tests/test_gpu_fullsize.py::test_bench_trajectory_int8_counts_widen_in_mid_run_at_full_size" \
  -x -v --timeout 300 --timeout-method thread > gpurun_out/r3/first_pytest.log 2>&1
rc=$?
tail -15 gpurun_out/r3/first_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3/bench_n1.json 2> gpurun_out/r3/bench_n1.err || exit $?
cat gpurun_out/r3/bench_n1.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); d.pop('cpu_baseline',None); print(json.dumps({k:d[k] for k in ('value','ms_per_step','objective_check','secondary')}, indent=1))"
timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r3/bench_two_ranks_one_gpu.json 2> gpurun_out/r3/bench_two_ranks_one_gpu.err
rc=$?
echo "two ranks rc=$rc"; tail -c 1500 gpurun_out/r3/bench_two_ranks_one_gpu.json; tail -5 gpurun_out/r3/bench_two_ranks_one_gpu.err
exit $rc
