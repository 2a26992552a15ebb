#!/bin/bash
mkdir -p gpurun_out/r04/pool
( while true; do date >> gpurun_out/r04/pool/heartbeat; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python -u -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py tests/test_gpu_parity.py -q -x -k "small or uniprot or config4 or switch or float16 or batch or view or packed" > gpurun_out/r04/pool/tests.log 2>&1; tail -3 gpurun_out/r04/pool/tests.log
grep -q "failed\|error" gpurun_out/r04/pool/tests.log && exit 1
for round in 1 2; do
  for w in 8 1; do python tools/c4_w8_time.py $w 2>&1 | grep world; done
done
R=$(pwd); OUT=$R/gpurun_out/r04/pool/prof; rm -rf $OUT; mkdir -p $OUT; PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4 --output-format csv -- $PY $R/tools/c4_share_trace.py 1 > $OUT/config4_world1.log 2>&1
cd $R; head -4 $OUT/kt_config4/*/*_kernel_stats.csv | cut -c1-150
