#!/bin/bash
python -m pytest tests -m gpu -x -q -k "fir or chain or config" 2>&1 | tail -3
for w in 4 5 16; do echo "== WPB=$w"; COMMS_OS1024_WPB=$w ALGOS=os1024 timeout -k 5 120 python scripts/bench_fir.py 255 24 200 2>&1 | grep -v amdgpu.ids; done
