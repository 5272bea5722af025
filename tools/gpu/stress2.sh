# randomised parity sweeps under the knobs of the one-launch iteration (each 300 random cases against the oracle)
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
: > gpurun_out/stress2.log
run() { echo "== $*" | tee -a gpurun_out/stress2.log; env "$@" timeout -k 10 400 python -u tools/stress_parity.py 300 "$seed" >> gpurun_out/stress2.log 2>&1; rc=$?; tail -1 gpurun_out/stress2.log; [ $rc -ne 0 ] && exit $rc; return 0; }
seed=31; run ADMM_DIRECT_LABELS=1
seed=32; run ADMM_DIRECT_LABELS=0 ADMM_DEFER_SUMS=0
seed=33; run ADMM_DIRECT_LABELS=1 ADMM_SWEEP_XCD=1 ADMM_POISON=171
seed=34; run ADMM_NOP=1
