#!/bin/bash
python -m pytest tests -m gpu -x -q -k "fir or chain" 2>&1 | tail -2
ALGOS=os1024 timeout -k 5 120 python scripts/bench_fir.py 255 24 300 2>&1 | grep -v amdgpu.ids
ALGOS=os1024 timeout -k 5 120 python scripts/bench_fir.py 255 26 100 2>&1 | grep -v amdgpu.ids
