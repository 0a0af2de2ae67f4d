#!/bin/bash
mkdir -p gpurun_out/tune
{
for extra in "" "-Xclang -target-feature -Xclang -load-store-opt"; do
  echo "=== EXTRA='$extra'"
  make -C comms_rs_amd/csrc -s clean; make -C comms_rs_amd/csrc -s -j16 EXTRA="$extra" 2>&1 | grep -E "error" 
  for w in 4 16; do
    echo "== WPB=$w"; COMMS_OS1024_WPB=$w ALGOS=os1024,os4096 timeout -k 5 120 python scripts/bench_fir.py 255 24 200 2>&1 | grep -v amdgpu.ids
  done
done
} 2>&1 | tee gpurun_out/tune/tune6.log
