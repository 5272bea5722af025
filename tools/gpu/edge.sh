# Edge solve: tests of the solver, then the rate of the bench workloads.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests/test_gpu_edge_exp.py tests/test_gpu_parity.py tests/test_gpu_edge_many_labels.py tests/test_gpu_edge_full_size.py tests/test_gpu_edge_ranks.py tests/test_gpu_graphfile.py tests/test_edge_log.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3/edge_pytest.log 2>&1
rc=$?; tail -6 gpurun_out/r3/edge_pytest.log; [ $rc -ne 0 ] && exit $rc
for w in edge_512x512_k32_potts edge_512x512_k16_trunc_linear edge_64x64_k4_potts; do
  for rep in 1 2; do
    timeout -k 10 300 python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3/edge_$w.json 2> gpurun_out/r3/edge_$w.err || exit $?
    python3 -c "import json,sys; d=json.load(open('gpurun_out/r3/edge_$w.json')); print('$w', '%.1f it/s'%d['value'], 'kernel %.4f ms'%d['roofline']['ms_per_launch'], d['final_objective']['linear'])"
  done
done
if [ -f tools/stress_edge.py ]; then timeout -k 10 600 python3 tools/stress_edge.py 300 2>&1 | tail -3; fi
