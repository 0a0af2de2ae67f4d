#!/bin/bash
python -m pytest tests -m gpu -x -q -k "fir or pulse or config1 or golden" 2>&1 | tail -3
for t in 16 32 63 64 127 255; do ALGOS=direct,os1024 timeout -k 5 120 python scripts/bench_fir.py $t 24 100 2>&1 | grep -v amdgpu.ids; done
