#!/bin/bash
# randomised runs of the many-small-alignments batch (result objects / view / view without traceback) and of the general generator:
# usage: r04_stress.sh [seed ...]   (each seed: 110 s of small batches, then 100 s of the general generator with seed + 1000)
mkdir -p gpurun_out/r04/stress
( while true; do date >> gpurun_out/r04/stress/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
for seed in "${@:-11 22}"; do
  timeout -k 10 200 python tests/stress_small_batches.py 110 $seed > gpurun_out/r04/stress/small_$seed.log 2>&1; tail -1 gpurun_out/r04/stress/small_$seed.log
  timeout -k 10 200 python tests/stress.py 100 $((seed + 1000)) > gpurun_out/r04/stress/general_$seed.log 2>&1; tail -1 gpurun_out/r04/stress/general_$seed.log
done
