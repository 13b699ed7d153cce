#!/bin/bash
# Runs GPU steps one after another on the GPU box; a step that times out (or is killed) ends
# the whole sequence -- no further GPU step is started behind a hung one.
# usage (inside the gpurun command):  source tools/gpu_steps.sh; step SECONDS command...
step() {
    local limit=$1; shift
    echo "[step] $*" >&2
    timeout -k 10 "$limit" "$@"
    local rc=$?
    echo "[step] rc=$rc" >&2
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "[step] timed out / killed: stopping here" >&2
        exit $rc
    fi
    return 0
}
