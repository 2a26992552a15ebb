#!/bin/bash
# checkpoints every 32 steps, guard 32: parity, then config 4 timings and kernel stats
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_fullsize.py tests/test_gpu_round4.py tests/test_gpu_stress.py -q -x -k "small_alignment or fullsize or config4 or switch or small_batches or uniprot" 2>&1 | tail -8 > gpurun_out/r04/t12.log; cat gpurun_out/r04/t12.log
grep -q "failed\|error" gpurun_out/r04/t12.log && exit 1
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/c4_eighth.json 2> gpurun_out/r04/c4_eighth.err; tail -c 500 gpurun_out/r04/c4_eighth.json
R=$(pwd); OUT=$R/gpurun_out/r04/prof12; mkdir -p $OUT; PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4 --output-format csv -- $PY $R/tools/c4_share_trace.py 1 > $OUT/config4_world1.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4_w8 --output-format csv -- $PY $R/tools/c4_share_trace.py 8 > $OUT/config4_world8.log 2>&1
cd $R
head -6 $OUT/kt_config4/*/*_kernel_stats.csv | cut -c1-170
grep -h "mi355_sw\|world" $OUT/config4_world1.log
