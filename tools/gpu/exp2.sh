set -u
cd "${GRAFT_REPO_ROOT:-.}"
for d in 0 1; do for x in 0 1; do echo "DIRECT=$d XCD=$x"; ADMM_DIRECT_LABELS=$d ADMM_SWEEP_XCD=$x timeout -k 10 200 python tools/gpu/exp_overhead.py 2048,512,64,batch,dense 200 || exit 1; done; done
