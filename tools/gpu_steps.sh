# tools/gpu_steps.sh — helper for multi-step gpurun calls: `step SECONDS LOGFILE cmd...` runs one step under its own time limit and
# STOPS the whole call if that step was killed at its limit (a GPU step that timed out: start no further GPU step in the same call);
# an ordinary failure (a failing assertion) is reported and the call goes on.  Source it: . tools/gpu_steps.sh
export TMPDIR=/tmp
mkdir -p gpurun_out
step() {
    local limit=$1 log=$2; shift 2
    echo "=== $(date +%T) $* (limit ${limit}s) -> $log"
    timeout -k 10 "$limit" "$@" > "$log" 2>&1
    local rc=$?
    echo "=== rc=$rc  $(tail -n 2 "$log" | cut -c1-300 | tr '\n' ' ')"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== step killed at its limit: stopping"; exit 1; fi
    return 0
}
