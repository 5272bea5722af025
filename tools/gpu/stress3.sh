# randomised parity sweeps (300 random cases each against the oracle) with one lane per label forced, so that small
# problems run the kernels that read a step's messages in pieces (32 and 64 labels), and with the column-pair placement off
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
: > gpurun_out/stress3.log
run() { echo "== $*" | tee -a gpurun_out/stress3.log; env "$@" timeout -k 10 500 python -u tools/stress_parity.py 300 "$seed" >> gpurun_out/stress3.log 2>&1; rc=$?; tail -1 gpurun_out/stress3.log; [ $rc -ne 0 ] && exit $rc; return 0; }
seed=41; run ADMM_SWEEP_NS=1
seed=42; run ADMM_SWEEP_NS=1 ADMM_SWEEP_POTTS=0 ADMM_POISON=171
seed=43; run ADMM_SWEEP_PAIRXCD=0 ADMM_SWEEP_TWOWAVE=0
seed=44; run ADMM_SWEEP_PAIRXCD=1 ADMM_SWEEP_TWOWAVE=0 ADMM_POISON=85
