#!/bin/bash
mkdir -p gpurun_out/r04/check
( while true; do date >> gpurun_out/r04/check/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -u -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_stress.py -q -x -k "small or uniprot or config4 or float16 or view or packed or small_batches" > gpurun_out/r04/check/tests.log 2>&1; tail -3 gpurun_out/r04/check/tests.log
grep -q "failed\|error" gpurun_out/r04/check/tests.log && exit 1
for round in 1 2 3; do for w in 1 8; do python tools/c4_w8_time.py $w 2>&1 | grep world; done; done
for i in 1 2 3; do python tools/c4_share_trace.py 1 0 2>&1 | grep "results on"; done
python tools/c4_share_trace.py 1 1 2>&1 | grep "results on"
