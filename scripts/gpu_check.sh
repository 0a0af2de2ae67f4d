#!/bin/bash
# One GPU-box visit: parity tests, smoke, a short bench, and a rocprofv3 kernel trace.
# Usage (from the repo root, on the GPU box):  bash scripts/gpu_check.sh [tag] [pytest -k filter]
set -o pipefail
TAG=${1:-run}
FILTER=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $OUT/summary.txt
if [ -n "$FILTER" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$FILTER" > $OUT/pytest.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
fi
rc=$?
tail -n 25 $OUT/pytest.log | tee -a $OUT/summary.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out: stopping" | tee -a $OUT/summary.txt; exit 1; fi
echo "== smoke" | tee -a $OUT/summary.txt
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -n 3 | tee -a $OUT/summary.txt
rc=${PIPESTATUS[0]}
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "smoke timed out: stopping"; exit 1; fi
echo "== bench" | tee -a $OUT/summary.txt
timeout -k 10 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
rc=$?
tail -n 5 $OUT/bench.err | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "bench timed out: stopping"; exit 1; fi
echo "== rocprofv3 kernel trace of the bench" | tee -a $OUT/summary.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
tail -n 3 $OUT/prof.err | tee -a $OUT/summary.txt
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -r head -n 12 | tee -a $OUT/summary.txt
