# randomised parity sweeps under the kernel-choice knobs (each 400 random cases against the oracle)
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
: > gpurun_out/stress.log
run() { echo "== $*" | tee -a gpurun_out/stress.log; env "$@" timeout -k 10 400 python -u tools/stress_parity.py 400 "$seed" >> gpurun_out/stress.log 2>&1; rc=$?; tail -1 gpurun_out/stress.log; [ $rc -ne 0 ] && exit $rc; return 0; }
seed=1;  run ADMM_NOP=1
seed=2;  run ADMM_SWEEP_TRUNC=1
seed=3;  run ADMM_SWEEP_TRUNC=1 ADMM_SWEEP_TWOWAVE=0
seed=5;  run ADMM_SWEEP_TWOWAVE=0
seed=7;  run ADMM_SWEEP_PCOMP=1 ADMM_SWEEP_TWOWAVE=0
seed=77; run ADMM_SWEEP_TWOWAVE=1 ADMM_LAM8=0
seed=11; run ADMM_SWEEP_TRUNC=1 ADMM_LAM16=0 ADMM_SWEEP_TWOWAVE=1
