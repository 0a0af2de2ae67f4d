#!/bin/bash
mkdir -p gpurun_out/tune
{
python -m pytest tests -m gpu -x -q -k "fir or smoke" 2>&1 | tail -4
for w in 4 16; do for r in 1 2 4; do
  echo "== WPB=$w MINRUN=$r"; COMMS_OS1024_MINRUN=$r COMMS_OS1024_WPB=$w ALGOS=os1024 timeout -k 5 120 python scripts/bench_fir.py 255 24 200 2>&1 | grep -v amdgpu.ids
done; done
ALGOS=os4096 timeout -k 5 120 python scripts/bench_fir.py 255 24 200 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/tune/tune4.log
