#!/bin/bash
# Run GPU steps one after another; an assertion failure (small exit code) lets the next step run, a timeout / kill
# (exit >= 124) stops the call - no further GPU step after a hung one.
# usage: tools/gpu_steps.sh "<cmd1>" "<cmd2>" ...
mkdir -p gpurun_out
final=0
for c in "$@"; do
    echo "=== $c" >&2
    bash -o pipefail -c "$c"
    rc=$?
    echo "=== rc=$rc" >&2
    if [ $rc -ge 124 ]; then exit $rc; fi
    if [ $rc -ne 0 ]; then final=$rc; fi
done
exit $final
