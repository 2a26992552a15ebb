#!/bin/bash
mkdir -p gpurun_out/r04/check
( while true; do date >> gpurun_out/r04/check/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -u -m pytest tests/test_gpu_stress.py tests/test_gpu_round4.py -q -x > gpurun_out/r04/check/tests.log 2>&1; tail -5 gpurun_out/r04/check/tests.log
