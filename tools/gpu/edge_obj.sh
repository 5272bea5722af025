#!/bin/bash
# Edge solver: parity tests, then the two edge workloads with the objective summed by the solve
# and (ADMM_EDGE_FUSED_OBJ=0) by the separate pass.
set -e
mkdir -p gpurun_out/edge
timeout -k 10 500 python -u -m pytest tests/test_edge_log.py tests/test_gpu_parity.py tests/test_gpu_edge_ranks.py tests/test_gpu_host_driver.py -m gpu -x -q --timeout 240 --timeout-method thread -k "edge or Edge or host or device" > gpurun_out/edge/tests.log 2>&1
tail -3 gpurun_out/edge/tests.log
for w in edge_512x512_k32_potts edge_512x512_k16_trunc_linear edge_64x64_k4_potts; do
  for f in 1 0; do
    ADMM_EDGE_FUSED_OBJ=$f timeout -k 10 200 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/edge/${w}_fused$f.json 2> gpurun_out/edge/${w}_fused$f.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/edge/${w}_fused$f.json").read().strip().splitlines()[-1])
print("$w fused=$f", d["value"], d["unit"], d["ms_per_step"], d.get("roofline",{}).get("kernels"))
PY
  done
done
