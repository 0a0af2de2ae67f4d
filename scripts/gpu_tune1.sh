#!/bin/bash
python -m pytest tests -m gpu -x -q -k "fir" 2>&1 | tail -3
ALGOS=os4096 timeout -k 5 120 python scripts/bench_fir.py 4097 26 20 2>&1 | grep -v amdgpu.ids
ALGOS=os4096 timeout -k 5 120 python scripts/bench_fir.py 1025 26 20 2>&1 | grep -v amdgpu.ids
ALGOS=os4096 timeout -k 5 120 python scripts/bench_fir.py 255 24 100 2>&1 | grep -v amdgpu.ids
