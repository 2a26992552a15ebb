#!/bin/bash
mkdir -p gpurun_out/r04/pool
( while true; do date >> gpurun_out/r04/pool/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -u -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_parity.py -q -x -k "small or uniprot or config4 or switch or float16 or batch or view or packed" > gpurun_out/r04/pool/tests.log 2>&1; tail -3 gpurun_out/r04/pool/tests.log
grep -q "failed\|error" gpurun_out/r04/pool/tests.log && exit 1
for round in 1 2; do
  for w in 8 4 2 1; do python tools/c4_w8_time.py $w 2>&1 | grep world; done
done
