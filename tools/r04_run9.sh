#!/bin/bash
# f16 first pass of the small-alignment batch: parity, then config 4 timings

mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py -q -x -k "small_alignment_batch_windows or every_switch" 2>&1 | tail -15 > gpurun_out/r04/t9a.log; cat gpurun_out/r04/t9a.log
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_round4.py -q -x 2>&1 | tail -15 > gpurun_out/r04/t9b.log; cat gpurun_out/r04/t9b.log
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/c4_sixth.json 2> gpurun_out/r04/c4_sixth.err; tail -c 1500 gpurun_out/r04/c4_sixth.json
R=$(pwd); OUT=$R/gpurun_out/r04/prof9; mkdir -p $OUT; PY=$(python3 -c 'import sys; print(sys.executable)')
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4 --output-format csv -- $PY $R/tools/c4_share_trace.py 1 > $OUT/config4_world1.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4_w8 --output-format csv -- $PY $R/tools/c4_share_trace.py 8 > $OUT/config4_world8.log 2>&1
cd $R
head -8 $OUT/kt_config4/*/*_kernel_stats.csv | cut -c1-160
tail -3 $OUT/config4_world1.log; tail -3 $OUT/config4_world8.log
