#!/bin/bash
# Run a list of GPU steps one after the other on the GPU box; every step under its own timeout, its
# output under gpurun_out/<tag>/<name>.{out,err}.  An ordinary failure (a red test, an assertion) is
# recorded and the next step still runs; a step that timed out or was killed ends the visit at once
# (no further GPU work after a hang).  Steps file: one per line, "name|timeout_seconds|command".
# Every step runs under `bash -o pipefail`: a red `pytest ... | tail` step is recorded red.
# Usage: bash scripts/gpu_steps.sh <tag> <steps-file>
set -o pipefail
TAG=$1
STEPS=$2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
fail=0
while IFS='|' read -r name tmo cmd; do
  [ -z "$name" ] && continue
  case "$name" in \#*) continue;; esac
  echo "== $name: $cmd" | tee -a $OUT/summary.txt
  t0=$(date +%s)
  timeout -k 10 "$tmo" bash -o pipefail -c "$cmd" > $OUT/$name.out 2> $OUT/$name.err < /dev/null
  rc=$?
  echo "   rc=$rc  $(( $(date +%s) - t0 )) s" | tee -a $OUT/summary.txt
  tail -n 6 $OUT/$name.out | cut -c1-600 | tee -a $OUT/summary.txt
  if [ $rc -ne 0 ]; then tail -n 12 $OUT/$name.err | cut -c1-400 | tee -a $OUT/summary.txt; fail=1; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || { [ $rc -ge 128 ] && [ $rc -ne 141 ]; }; then
    echo "step $name timed out or was killed: stopping the visit" | tee -a $OUT/summary.txt
    exit 1
  fi
done < "$STEPS"
exit $fail
