# randomised parity sweeps (400 random cases each against the oracle) on the library's own choices and with the
# two-wavefront kernels forced on and off, other seeds than the earlier sweeps
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
: > gpurun_out/stress4.log
run() { echo "== $*" | tee -a gpurun_out/stress4.log; env "$@" timeout -k 10 700 python -u tools/stress_parity.py 400 "$seed" >> gpurun_out/stress4.log 2>&1; rc=$?; tail -1 gpurun_out/stress4.log; [ $rc -ne 0 ] && exit $rc; return 0; }
seed=51; run ADMM_POISON=17
seed=52; run ADMM_SWEEP_TWOWAVE=1
seed=53; run ADMM_SWEEP_TWOWAVE=0 ADMM_DIRECT_LABELS=0
seed=54; run ADMM_SWEEP_TRUNC=1 ADMM_LAM8_THR=2 ADMM_LAM8_EVERY=2
seed=55; run ADMM_SWEEP_PCOMP=1 ADMM_DIRECT_LABELS=1
