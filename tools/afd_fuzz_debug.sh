#!/bin/bash
# AFD fuzz with details of the first mismatching scenarios (run on the GPU box): usage afd_fuzz_debug.sh [seeds...]
mkdir -p gpurun_out
for seed in ${@:-1 5}; do
  FUZZ_AFD=1 python tools/fuzz_scenarios.py 60 $seed > gpurun_out/afd_fuzz_$seed.log 2>&1
  grep -E "^MISMATCH|scenarios run" gpurun_out/afd_fuzz_$seed.log | head -40
  for it in $(grep -E "^MISMATCH" gpurun_out/afd_fuzz_$seed.log | awk '{print $2}' | head -6); do
    FUZZ_AFD=1 python tools/fuzz_scenarios.py 60 $seed $it > gpurun_out/afd_fuzz_${seed}_$it.log 2>&1
  done
done
