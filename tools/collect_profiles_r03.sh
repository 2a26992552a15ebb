#!/usr/bin/env bash
# Round 3 measurement records on the GPU box -> gpurun_out/profiles_r03/ (copied to profiles/ afterwards): kernel-trace stats of the
# bench command, PMC passes (separate runs, no trace flags) for the headline kernel AND the kernels the round-2 review found
# unprofiled: sw_wave_kernel (config 4), sw_long_kernel / sw_strip_kernel / walk (config 5), sw_solo_kernel (one-by-one calls).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_r03
mkdir -p $OUT $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
B="$PY $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-strong --no-traffic --no-parity"
rocprofv3 --kernel-trace --stats -d $OUT/kt_bench --output-format csv -- $B > $OUT/kt_bench.json 2> $OUT/kt_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt_config4 --output-format csv -- $PY $R/tools/c4_packed.py > $OUT/config4.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_config5 --output-format csv -- $PY $R/tools/c5_whole.py > $OUT/config5.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_one_by_one --output-format csv -- $PY $R/tools/lat_probe.py 50000000 0 > $OUT/one_by_one.log 2>&1
cd $R
tools/pmc_run.sh bench 'sw_score_kernel' python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-strong --no-traffic --no-parity > $OUT/pmc_bench.log 2>&1
tools/pmc_run.sh config4 'sw_wave_prof_kernel<10, true, true>|sw_wave_prof_kernel<10, true, false>|sw_wave_walk_kernel' python3 $R/tools/c4_packed.py > $OUT/pmc_config4.log 2>&1
tools/pmc_run.sh config5 'sw_long_kernel|sw_strip_kernel<3, false, 2|sw_strip_kernel<3, false, 0|sw_wave_walk_long_kernel' python3 $R/tools/c5_whole.py > $OUT/pmc_config5.log 2>&1
tools/pmc_run.sh one_by_one 'sw_solo_kernel|sw_score_kernel' python3 $R/tools/lat_probe.py 50000000 0 > $OUT/pmc_one_by_one.log 2>&1
cp gpurun_out/pmc/*.json $OUT/ 2>/dev/null
find $OUT -name "*kernel_stats.csv" | head; ls $OUT
