#!/bin/bash
for w in 2 3 4; do echo "== WPS=$w"; COMMS_OS4096_WPS=$w ALGOS=os4096 timeout -k 5 120 python scripts/bench_fir.py 255 24 100 2>&1 | grep -v amdgpu.ids;  COMMS_OS4096_WPS=$w ALGOS=os4096 timeout -k 5 120 python scripts/bench_fir.py 1025 26 20 2>&1 | grep -v amdgpu.ids; done
