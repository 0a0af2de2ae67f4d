python -m pytest tests -m gpu -x -q -k "fft" 2>&1 | tail -2
timeout -k 5 120 python scripts/bench_fft.py 10 20 2>&1 | grep -v amdgpu.ids
