#!/usr/bin/env bash
# Round 4 measurement records on the GPU box -> gpurun_out/profiles_r04/ (copied to profiles/ afterwards): kernel-trace stats of the
# bench command and of configs 4 / 5 (whole job and one rank's share at world 8), PMC passes (separate runs, no trace flags) for
# the headline kernel, sw_long_kernel WITH its saved state, and the block kernels of the finish.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_r04
mkdir -p $OUT $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
B="$PY $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-strong --no-traffic --no-parity"
rocprofv3 --kernel-trace --stats -d $OUT/kt_bench --output-format csv -- $B > $OUT/kt_bench.json 2> $OUT/kt_bench.err
echo "kt_bench rc=$?"
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4 --output-format csv -- $PY $R/tools/c4_share_trace.py 1 > $OUT/config4_world1.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4_w8 --output-format csv -- $PY $R/tools/c4_share_trace.py 8 > $OUT/config4_world8.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_config5 --output-format csv -- $PY $R/tools/c5_whole.py > $OUT/config5.log 2>&1
echo "kt done"
cd $R
tools/pmc_run.sh bench 'sw_score_kernel' python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-strong --no-traffic --no-parity > $OUT/pmc_bench.log 2>&1
tools/pmc_run.sh config5 'sw_long_kernel|sw_strip_kernel<5, false, 2|sw_strip_kernel<5, false, 0|sw_wave_walk_long_kernel' python3 $R/tools/c5_whole.py > $OUT/pmc_config5.log 2>&1
tools/pmc_run.sh config5_nosave 'sw_long_kernel' python3 $R/tools/c5_whole.py no_long_save > $OUT/pmc_config5_nosave.log 2>&1
cp gpurun_out/pmc/*.json $OUT/ 2>/dev/null
find $OUT -name "*kernel_stats.csv" | head; ls $OUT
