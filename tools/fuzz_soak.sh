#!/bin/bash
# Soak run of the scenario fuzzer on the GPU box (round 6): seeds the test suite does not hold, every mode — posteriors / MAP, AFD
# lists, prior scenarios, many events (wide build).  One summary line per (mode, seed) -> gpurun_out/fuzz_soak.txt
cd "$(dirname "$0")/.."
O=gpurun_out/fuzz_soak.txt; : > $O
run() { tag=$1; shift; env "$@" python tools/fuzz_scenarios.py $N $SEED 2>&1 | tail -1 | sed "s/^/$tag seed $SEED: /" | tee -a $O; }
for SEED in 11 12 13 14 15 16; do N=60 run plain FUZZ_X=0; done
for SEED in 21 22 23 24; do N=60 run afd FUZZ_AFD=1; done
for SEED in 31 32 33; do N=40 run prior FUZZ_PRIOR=1; done
for SEED in 41 42; do N=40 run prior-afd FUZZ_PRIOR=1 FUZZ_AFD=1; done
for SEED in 51 52; do N=10 run many-events FUZZ_MANY_EVENTS=1 FUZZ_AFD=1 FUZZ_AFD_CAP=2048; done
for SEED in 61 62; do N=40 run types FUZZ_TYPES=1; done
