#!/bin/bash
# after the float16 first pass of the small-alignment batch: whole GPU suite, config 4 timings + host phases + kernel stats
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -q -x -m gpu --deselect tests/test_gpu_stress.py 2>&1 | tail -15 > gpurun_out/r04/t10.log; cat gpurun_out/r04/t10.log
grep -q " passed" gpurun_out/r04/t10.log || exit 1
grep -q "failed" gpurun_out/r04/t10.log && exit 1
timeout -k 10 300 python tools/run_config4_share.py > gpurun_out/r04/c4_seventh.json 2> gpurun_out/r04/c4_seventh.err; tail -c 600 gpurun_out/r04/c4_seventh.json
python tools/c4_share_trace.py 1 > gpurun_out/r04/c4_trace1_tb.log 2>&1; python tools/c4_share_trace.py 1 1 > gpurun_out/r04/c4_trace1_so.log 2>&1
python tools/c4_share_trace.py 8 > gpurun_out/r04/c4_trace8_tb.log 2>&1; python tools/c4_share_trace.py 8 1 > gpurun_out/r04/c4_trace8_so.log 2>&1
grep -h "mi355_sw\|world" gpurun_out/r04/c4_trace1_tb.log gpurun_out/r04/c4_trace1_so.log gpurun_out/r04/c4_trace8_so.log
R=$(pwd); OUT=$R/gpurun_out/r04/prof10; mkdir -p $OUT; PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4 --output-format csv -- $PY $R/tools/c4_share_trace.py 1 > $OUT/config4_world1.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4_w8 --output-format csv -- $PY $R/tools/c4_share_trace.py 8 > $OUT/config4_world8.log 2>&1
cd $R
head -6 $OUT/kt_config4/*/*_kernel_stats.csv | cut -c1-170
