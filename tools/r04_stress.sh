#!/bin/bash
# randomised runs of the many-small-alignments batch (result objects / view / view without traceback) and of the general generator
mkdir -p gpurun_out/r04/stress
( while true; do date >> gpurun_out/r04/stress/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
for seed in 11 22 33 44; do
  timeout -k 10 200 python tests/stress_small_batches.py 110 $seed > gpurun_out/r04/stress/small_$seed.log 2>&1; tail -1 gpurun_out/r04/stress/small_$seed.log
done
for seed in 55 66; do
  timeout -k 10 200 python tests/stress.py 100 $seed > gpurun_out/r04/stress/general_$seed.log 2>&1; tail -1 gpurun_out/r04/stress/general_$seed.log
done
