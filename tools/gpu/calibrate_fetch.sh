# FETCH_SIZE against known byte counts in the sweep's read patterns (tools/fetch_calibration.hip):
# one pass with the derived counter, one with the raw request counters it is made of.
set -u
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-$PWD}"
out="$root/gpurun_out/cal"; mkdir -p "$out"
cd /tmp
timeout -k 10 120 "$root/tools/fetch_calibration" > "$out/known_bytes.json" || exit $?
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o cal --output-format csv -- "$root/tools/fetch_calibration" > "$out/fetch.log" 2>&1 || exit $?
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d "$out/raw" -o cal --output-format csv -- "$root/tools/fetch_calibration" > "$out/raw.log" 2>&1 || exit $?
cat "$out/known_bytes.json"
